#!/bin/bash
# quick loop: primitives + parity + densify-loop tests (fail fast), then a bench line without the CPU baseline
R=$GRAFT_REPO_ROOT
TAG=${1:-q}
E=$R/gpurun_out/q_$TAG
mkdir -p $E
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_parity.py tests/test_gpu_densify_loop.py -x -q --tb=short -m gpu 2>&1 | grep -v amdgpu.ids | tail -25 | tee $E/tests.log
timeout 600 python bench.py --no-cpu-baseline ${@:2} 2>$E/bench.err | grep -v amdgpu.ids | tail -1 > $E/bench.json
python - <<PY
import json
b=json.load(open("$E/bench.json"))
print("it/s", b["value"], "ms", b["ms_per_step"], "sustained", b.get("sustained",{}).get("ms_per_step"))
print("stages", b["roofline"]["stages_ms"])
print("frac", b["roofline"]["stages_hbm_frac"])
for o in b.get("other_configs",[]):
    print(o["config"], o.get("ms_per_step", o.get("ms_per_step_amortised")), o.get("stages_ms"), o.get("error"), o.get("raster_ms_per_step"), o.get("densify_ms_mean"))
PY
tail -3 $E/bench.err
