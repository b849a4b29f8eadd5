#!/bin/bash
# round 5, call A: instruction costs in real cycles (tools/ubench/valu_rates2), then the sparse-visit path of the blend
# backward: correctness + determinism of the two forms, and an A/B of thresholds against the dense-only build.
R=$GRAFT_REPO_ROOT; E=$R/gpurun_out/r5_a; mkdir -p $E; cd $R
V=$R/street_gaussians_amd/variants
timeout 300 tools/ubench/valu_rates2 > $E/valu_rates2.jsonl 2> $E/ubench.err
SEL="backward_matches_oracle or deterministic or random_scenes or exact_mode_cases or edge_sizes"
for v in shipped m2k8; do
  if [ $v = shipped ]; then L=""; else L="SGR_LIB=$V/libsgr_hip_$v.so"; fi
  env SGR_BINDING=ctypes $L timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "$SEL" 2>&1 | grep -v amdgpu.ids | tail -4 > $E/pytest_$v.log
done
for v in shipped m2k8 m2k16; do
  if [ $v = shipped ]; then L=""; else L="SGR_LIB=$V/libsgr_hip_$v.so"; fi
  env SGR_BINDING=ctypes SOAK_CONFIGS=0 $L timeout 300 python tools/soak.py 60 2>&1 | grep -v amdgpu.ids | tail -2 > $E/soak_$v.log
done
run() {
  python $R/bench.py --no-cpu-baseline --no-other-configs --steps 200 --warmup 10 --device-warmup 0.5 2>/dev/null | tail -1 | python -c "
import sys,json; b=json.loads(sys.stdin.read()); pm=b.get('parity_mode') or {}
print(json.dumps({'variant':'$1','ms':b['ms_per_step'],'timed':b['timed_region']['ms_per_step'],'exact_ms':b.get('ms_per_step_exact'),'bwd':b['roofline']['stages_ms'].get('blend_bwd'),'fwd':b['roofline']['stages_ms'].get('blend_fwd'),'exact_bwd':pm.get('blend_bwd_ms'),'exact_fwd':pm.get('blend_fwd_ms'),'kernel_ms':b['roofline']['kernel_ms']}))"
}
for rep in 1 2; do
  SGR_BINDING=ctypes run shipped_m1k8 >> $E/ab.jsonl
  for v in off m1k4 m2k4 m2k8 m2k12 m2k16 m2k24; do SGR_BINDING=ctypes SGR_LIB=$V/libsgr_hip_$v.so run $v >> $E/ab.jsonl; done
done
cat $E/pytest_*.log $E/soak_*.log $E/ab.jsonl
