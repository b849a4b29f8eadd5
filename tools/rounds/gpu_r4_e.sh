#!/bin/bash
# round 4, call E: branch-free scalar walk -- staggered prefetch (shipped), both visits in one region + late prefetch (swpf2),
# no prefetch (swnopf), against the LDS kernel
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/r4e
mkdir -p $E
cd $R
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_parity.py -x -q --tb=short -m gpu 2>&1 | grep -v amdgpu.ids | tail -5 | tee $E/pytest_parity.log
run() {  # tag, extra bench args
  python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other-configs $2 2>/dev/null | tail -1 > $E/bench_$1.json
  python - <<PY
import json
b = json.load(open("$E/bench_$1.json"))
print("$1", "value", b["value"], "ms", b["ms_per_step"], "exact", b.get("value_exact"), b.get("ms_per_step_exact"), {k: (b.get("parity_mode") or {}).get(k) for k in ("blend_fwd_ms", "blend_bwd_ms", "gauss_bwd_ms")}, {k: b["roofline"]["stages_ms"][k] for k in ("blend_fwd", "blend_bwd", "gauss_bwd")})
PY
}
export SGR_BINDING=ctypes
for rep in 1 2; do
  run sw_$rep ""
  SGR_LIB=$R/street_gaussians_amd/variants/libsgr_hip_swpf2.so run swpf2_$rep ""
  SGR_LIB=$R/street_gaussians_amd/variants/libsgr_hip_swnopf.so run swnopf_$rep ""
  SGR_NO_SW=1 run lds_$rep ""
done
