#!/bin/bash
# round 5, call F: (1) where the 20-step timed region loses time against the sustained one (per-step GPU marks);
# (2) the scan's last step folded into the duplicate kernel: parity + primitives tests, stage times at 1 M and 5 M
R=$GRAFT_REPO_ROOT; E=$R/gpurun_out/r5_f; mkdir -p $E; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_primitives.py tests/test_gpu_callsite.py -q -x -m gpu 2>&1 | grep -v amdgpu.ids | tail -5 > $E/pytest.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -m gpu -k "test_baseline_size_matches_oracle and (headline_1M or 5M)" 2>&1 | grep -v amdgpu.ids | grep -v "^{" | tail -5 >> $E/pytest.log
for i in 1 2 3; do
  SGR_BENCH_REGION_TRACE=1 python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 2> $E/trace_$i.err | tail -1 | python -c "
import sys,json; b=json.loads(sys.stdin.read()); print(json.dumps({'ms':b['ms_per_step'],'sustained':(b.get('sustained') or {}).get('ms_per_step'),'stages':b['roofline']['stages_ms']}))" >> $E/region.jsonl
  grep region-trace $E/trace_$i.err | head -3 >> $E/region.jsonl
done
python bench.py --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | tail -1 > $E/bench.json
python - <<PY >> $E/region.jsonl
import json
b = json.load(open("$E/bench.json"))
print(json.dumps(b["summary"]))
for c in b.get("other_configs", []):
    print(c.get("config"), c.get("ms_per_step"), c.get("stages_ms"), c.get("ms_per_step_amortised"))
PY
cat $E/pytest.log $E/region.jsonl
