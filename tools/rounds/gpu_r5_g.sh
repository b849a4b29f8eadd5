#!/bin/bash
R=$GRAFT_REPO_ROOT; E=$R/gpurun_out/r5_g; mkdir -p $E; cd $R
for i in 1 2; do
  SGR_BENCH_REGION_TRACE=1 python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 2> $E/trace_$i.err | tail -1 > $E/b_$i.json
  grep region-trace $E/trace_$i.err | head -6
done
SGR_BENCH_REGION_TRACE=1 python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 --device-warmup 0 2> $E/trace_3.err | tail -1 > $E/b_3.json
grep region-trace $E/trace_3.err | head -6
timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $E/bench.json
python - <<PY
import json
b = json.load(open("$E/bench.json"))
print(json.dumps(b["summary"]))
for c in b.get("other_configs", []):
    print(c.get("config"), c.get("ms_per_step"), c.get("stages_ms"), c.get("ms_per_step_amortised"), c.get("host_ms_to_queue_one_iteration"), c.get("host_ms_in_step_call_incl_gpu_wait"), c.get("pool_bytes_reserved"), (c.get("allocator") or {}).get("reserved_bytes.all.peak"), c.get("device_allocations_in_region"), c.get("other_run"))
PY
