#!/bin/bash
# round 5, call B: the visit-row blend backward (plain LDS stores instead of ds_add_f32) + sparse-visit thresholds:
# parity suite + 1M full-size tests (incl. the new entry-for-entry leg and the independent cut check) on the shipped build,
# LDS instruction costs by active lanes, then the A/B against the round-4 combine.
R=$GRAFT_REPO_ROOT; E=$R/gpurun_out/r5_c; mkdir -p $E; cd $R
V=$R/street_gaussians_amd/variants
timeout 300 tools/ubench/valu_rates2 quick > $E/valu_rates2.jsonl 2> $E/ubench.err
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_primitives.py -q -x -m gpu 2>&1 | grep -v amdgpu.ids | tail -15 > $E/pytest_parity.log
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -x -m gpu -k "headline_1M" 2>&1 | grep -v amdgpu.ids | grep -v "^{" | tail -15 > $E/pytest_fullsize_1M.log
cp gpurun_out/fullsize_parity.json gpurun_out/threeway_fullsize.json $E/ 2>/dev/null
env SOAK_CONFIGS=0,1,2 timeout 600 python tools/soak.py 60 2>&1 | grep -v amdgpu.ids | tail -4 > $E/soak.log
run() {
  python $R/bench.py --no-cpu-baseline --no-other-configs --steps 200 --warmup 10 --device-warmup 0.5 2>/dev/null | tail -1 | python -c "
import sys,json; b=json.loads(sys.stdin.read()); m=b.get('modes') or {}; ex=m.get('exact') or {}; sx=m.get('strict') or {}
print(json.dumps({'variant':'$1','ms':b['ms_per_step'],'exact_ms':b.get('ms_per_step_exact'),'strict_ms':b.get('ms_per_step_strict'),'bwd':b['roofline']['stages_ms'].get('blend_bwd'),'fwd':b['roofline']['stages_ms'].get('blend_fwd'),'gauss':b['roofline']['stages_ms'].get('gauss_bwd'),'exact_bwd':(ex.get('stages_ms') or {}).get('blend_bwd'),'exact_fwd':(ex.get('stages_ms') or {}).get('blend_fwd'),'strict_bwd':(sx.get('stages_ms') or {}).get('blend_bwd'),'kernel_ms':b['roofline']['kernel_ms']}))"
}
for rep in 1 2; do
  SGR_BINDING=ctypes run shipped_vr_k8 >> $E/ab.jsonl
  for v in legacy legacyk4 vr_k0 vr_k4 vr_k6 vr_k9; do SGR_BINDING=ctypes SGR_LIB=$V/libsgr_hip_$v.so run $v >> $E/ab.jsonl; done
done
true
cat $E/pytest_*.log $E/soak.log $E/ab.jsonl
