#!/bin/bash
# round 4, call G: depth sort on 27 key bits in three 9-bit passes (shipped) vs four passes (SGR_SORT_BITS=8): primitives and
# parity suites, stage times at 1 M / 5 M / 500 k
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/r4g
mkdir -p $E
cd $R
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_parity.py tests/test_gpu_densify_loop.py -x -q --tb=short -m gpu 2>&1 | grep -v amdgpu.ids | tail -8 | tee $E/pytest.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q --tb=short -m gpu -k "baseline_size" 2>&1 | grep -v amdgpu.ids | grep -v "^{" | tail -4 | tee -a $E/pytest.log
run() {  # tag, extra bench args
  python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other-configs $2 2>/dev/null | tail -1 > $E/bench_$1.json
  python - <<PY
import json
b = json.load(open("$E/bench_$1.json"))
print("$1", "value", b["value"], "ms", b["ms_per_step"], "exact", b.get("ms_per_step_exact"), {k: b["roofline"]["stages_ms"][k] for k in ("preprocess", "scan", "sort", "duplicate")})
PY
}
for cfg in "1M:" "5M:--gaussians 5000000 --steps 30" "500k:--gaussians 500000"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  run p3_$tag "$args"
  SGR_SORT_BITS=8 run p4_$tag "$args"
  run p3b_$tag "$args"
done
