#!/bin/bash
# round 4, call M: per-Gaussian tile masks inside the cut-down rects (shipped) vs the bounding-box rects alone
# (SGR_NO_TILE_MASK=1): whole -m gpu suite, then stage times at 1 M / 5 M / 500 k / 2 M + 19 channels
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/r4m
mkdir -p $E
cd $R
rm -f gpurun_out/parity_measured.jsonl gpurun_out/threeway_fullsize.json gpurun_out/fullsize_parity.json
timeout 1800 python -m pytest tests -q --tb=short -m gpu 2>&1 | grep -v amdgpu.ids | grep -v "^{" | tail -30 | tee $E/pytest_gpu.log
run() {  # tag, extra bench args
  python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other-configs $2 2>/dev/null | tail -1 > $E/bench_$1.json
  python - <<PY
import json
b = json.load(open("$E/bench_$1.json"))
c = b["config"]
print("$1", "ms", b["ms_per_step"], "exact", b.get("ms_per_step_exact"), "R", c["num_rendered_R"], "emitted", c.get("instances_emitted"), {k: v for k, v in b["roofline"]["stages_ms"].items()})
PY
}
for cfg in "1M:" "5M:--gaussians 5000000 --steps 30" "500k:--gaussians 500000" "2M_S19:--gaussians 2000000 --semantics 19 --steps 30"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  run mask_$tag "$args"
  SGR_NO_TILE_MASK=1 run bbox_$tag "$args"
done
run mask2_1M ""
