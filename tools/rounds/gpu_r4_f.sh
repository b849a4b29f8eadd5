#!/bin/bash
# round 4, call F: wave-cooperative row sum (shipped) vs the four-lanes-per-Gaussian one (SGR_RS_QUADS=1): parity suites, then
# stage times at 500 k / 1 M / 5 M and 2 M + 19 channels
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/r4f
mkdir -p $E
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_densify_loop.py tests/test_gpu_callsite.py tests/test_gpu_scene.py -x -q --tb=short -m gpu 2>&1 | grep -v amdgpu.ids | tail -8 | tee $E/pytest.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q --tb=short -m gpu -k "baseline_size" 2>&1 | grep -v amdgpu.ids | grep -v "^{" | tail -4 | tee -a $E/pytest.log
run() {  # tag, extra bench args
  python $R/bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-other-configs $2 2>/dev/null | tail -1 > $E/bench_$1.json
  python - <<PY
import json
b = json.load(open("$E/bench_$1.json"))
print("$1", "value", b["value"], "ms", b["ms_per_step"], "exact", b.get("ms_per_step_exact"), {k: b["roofline"]["stages_ms"][k] for k in ("preprocess", "blend_fwd", "blend_bwd", "gauss_bwd")})
PY
}
for cfg in "1M:" "500k:--gaussians 500000" "5M:--gaussians 5000000 --steps 30" "2MS19:--gaussians 2000000 --semantics 19 --steps 30"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  run wave_$tag "$args"
  SGR_RS_QUADS=1 run quads_$tag "$args"
done
