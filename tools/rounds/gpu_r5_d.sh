#!/bin/bash
# round 5, call D: sparse-visit stage in the free tail of the row array (adaptive threshold up to 16): tests, thresholds
# A/B, and the other configurations (2 M + 19 channels, 5 M) against the round-4 combine.
R=$GRAFT_REPO_ROOT; E=$R/gpurun_out/r5_d; mkdir -p $E; cd $R
V=$R/street_gaussians_amd/variants
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_primitives.py -q -x -m gpu 2>&1 | grep -v amdgpu.ids | tail -8 > $E/pytest_parity.log
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -x -m gpu -k "headline_1M or 2M_S19" 2>&1 | grep -v amdgpu.ids | grep -v "^{" | tail -8 > $E/pytest_fullsize.log
cp gpurun_out/fullsize_parity.json gpurun_out/threeway_fullsize.json $E/ 2>/dev/null
env SOAK_CONFIGS=0,1,3 timeout 600 python tools/soak.py 100 2>&1 | grep -v amdgpu.ids | tail -4 > $E/soak.log
run() {
  python $R/bench.py --no-cpu-baseline --no-other-configs --steps 200 --warmup 10 --device-warmup 0.5 2>/dev/null | tail -1 | python -c "
import sys,json; b=json.loads(sys.stdin.read()); m=b.get('modes') or {}; ex=m.get('exact') or {}; sx=m.get('strict') or {}
print(json.dumps({'variant':'$1','ms':b['ms_per_step'],'exact_ms':b.get('ms_per_step_exact'),'strict_ms':b.get('ms_per_step_strict'),'bwd':b['roofline']['stages_ms'].get('blend_bwd'),'fwd':b['roofline']['stages_ms'].get('blend_fwd'),'gauss':b['roofline']['stages_ms'].get('gauss_bwd'),'exact_bwd':(ex.get('stages_ms') or {}).get('blend_bwd'),'exact_fwd':(ex.get('stages_ms') or {}).get('blend_fwd'),'strict_bwd':(sx.get('stages_ms') or {}).get('blend_bwd'),'kernel_ms':b['roofline']['kernel_ms']}))"
}
for rep in 1 2; do
  SGR_BINDING=ctypes run shipped_t_k16 >> $E/ab.jsonl
  for v in legacy t_k0 t_k8 t_k12; do SGR_BINDING=ctypes SGR_LIB=$V/libsgr_hip_$v.so run $v >> $E/ab.jsonl; done
done
oc() {
  python $R/bench.py --no-cpu-baseline --steps 50 --warmup 5 --device-warmup 0.5 2>/dev/null | tail -1 > $E/bench_$1.json
  python -c "
import json; b=json.load(open('$E/bench_$1.json'))
for c in b.get('other_configs', []):
    print('$1', c.get('config'), c.get('ms_per_step'), c.get('blend_bwd_ms'), c.get('blend_fwd_ms'), (c.get('stages_ms') or {}).get('gauss_bwd'), c.get('ms_per_step_amortised'))"
}
SGR_BINDING=ctypes oc shipped >> $E/other.txt
SGR_BINDING=ctypes SGR_LIB=$V/libsgr_hip_legacy.so oc legacy >> $E/other.txt
cat $E/pytest_*.log $E/soak.log $E/ab.jsonl $E/other.txt
