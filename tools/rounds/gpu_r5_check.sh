#!/bin/bash
# round 5 iteration loop: whole -m gpu suite (summary + failures), smoke(), one default bench line's headline figures
R=$GRAFT_REPO_ROOT; TAG=${1:-chk}; E=$R/gpurun_out/r5_$TAG; mkdir -p $E; cd $R
rm -f gpurun_out/parity_measured.jsonl gpurun_out/threeway_fullsize.json gpurun_out/fullsize_parity.json
timeout 2400 python -m pytest tests -q --tb=short -m gpu 2>&1 | grep -v amdgpu.ids | grep -v "^{" | tail -40 | tee $E/pytest_gpu.log
cp gpurun_out/parity_measured.jsonl gpurun_out/threeway_fullsize.json gpurun_out/fullsize_parity.json $E/ 2>/dev/null
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $E/smoke.log
timeout 900 python bench.py $2 2>/dev/null | tail -1 > $E/bench.json
python - <<PY
import json
b = json.load(open("$E/bench.json"))
print(json.dumps(b["summary"]))
print(b["roofline"]["stages_ms"])
for c in b.get("other_configs", []):
    print(c.get("config"), c.get("ms_per_step"), c.get("stages_ms"), c.get("ms_per_step_amortised"), c.get("host_ms_to_queue_one_iteration"))
PY
