#!/bin/bash
R=$GRAFT_REPO_ROOT; E=$R/gpurun_out/r5_l; mkdir -p $E; cd $R
timeout 600 python -m pytest tests/test_gpu_graph.py -q -x -m gpu --tb=short 2>&1 | grep -v amdgpu.ids | tail -25
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_primitives.py tests/test_gpu_densify_loop.py -q -x -m gpu --tb=short 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|Error" | tail -5
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 2>$E/bench.err | tail -1 > $E/bench.json
python - <<PY
import json
b = json.load(open("$E/bench.json"))
print(json.dumps(b["summary"])); print(b.get("lazy"))
PY
tail -3 $E/bench.err
