#!/bin/bash
# round 4, call K: forward blend with its three per-visit selects on an SGPR-pair mask (shipped) vs the compiler's VCC selects
# (variant sel0): parity suite of the forward, then stage times, both through the ctypes binding
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/r4k
mkdir -p $E
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q --tb=short -m gpu 2>&1 | grep -v amdgpu.ids | tail -5 | tee $E/pytest.log
run() {  # tag, extra bench args
  python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs $2 2>/dev/null | tail -1 > $E/bench_$1.json
  python - <<PY
import json
b = json.load(open("$E/bench_$1.json"))
print("$1", "ms", b["ms_per_step"], "exact", b.get("ms_per_step_exact"), {k: v for k, v in b["roofline"]["stages_ms"].items() if k in ("blend_fwd", "blend_bwd")}, "exact fwd", b["parity_mode"]["blend_fwd_ms"])
PY
}
export SGR_BINDING=ctypes
for i in 1 2; do
  run sel1_$i ""
  SGR_LIB=$R/street_gaussians_amd/variants/libsgr_hip_sel0.so run sel0_$i ""
done
run sel1_2M "--gaussians 2000000 --semantics 19 --steps 30"
SGR_LIB=$R/street_gaussians_amd/variants/libsgr_hip_sel0.so run sel0_2M "--gaussians 2000000 --semantics 19 --steps 30"
