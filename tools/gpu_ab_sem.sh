R=$GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_parity.py -x -q --tb=short -m gpu 2>&1 | grep -v amdgpu.ids | tail -4
for S in 3 8 12 22; do
  for v in shipped fwdold; do
    if [ $v = shipped ]; then L=""; else L="SGR_LIB=$R/street_gaussians_amd/variants/libsgr_hip_$v.so"; fi
    env SGR_BINDING=ctypes $L python $R/bench.py --no-cpu-baseline --no-other-configs --steps 100 --semantics $S 2>/dev/null | tail -1 | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('S=$S', '$v', b['ms_per_step'], 'fwd', b['roofline']['stages_ms']['blend_fwd'], 'bwd', b['roofline']['stages_ms']['blend_bwd'])"
  done
done
