#!/bin/bash
# round 4 iteration loop: whole -m gpu suite (summary + failures), then one bench line's headline figures
R=$GRAFT_REPO_ROOT
TAG=${1:-chk}
E=$R/gpurun_out/r4_$TAG
mkdir -p $E
cd $R
rm -f gpurun_out/parity_measured.jsonl gpurun_out/threeway_fullsize.json gpurun_out/fullsize_parity.json
timeout 1800 python -m pytest tests -q --tb=short -m gpu 2>&1 | grep -v amdgpu.ids | grep -v "^{" | tail -40 | tee $E/pytest_gpu.log
cp gpurun_out/parity_measured.jsonl gpurun_out/threeway_fullsize.json gpurun_out/fullsize_parity.json $E/ 2>/dev/null
timeout 900 python bench.py --no-cpu-baseline --no-other-configs $2 2>/dev/null | tail -1 > $E/bench.json
python - <<PY
import json
b = json.load(open("$E/bench.json"))
print("value", b["value"], "ms", b["ms_per_step"], "exact", b.get("value_exact"), b.get("ms_per_step_exact"), "timed", b["timed_region"])
print("roofline frac", b["roofline"]["frac"], "valu_issue", b["roofline"].get("valu_issue"))
print(b["roofline"]["stages_ms"])
PY
