#!/bin/bash
R=$GRAFT_REPO_ROOT; E=$R/gpurun_out/r5_h; mkdir -p $E; cd $R
timeout 600 python -m pytest tests/test_gpu_densify.py tests/test_gpu_densify_loop.py tests/test_gpu_parity.py -q -x -m gpu 2>&1 | grep -v amdgpu.ids | tail -4
SGR_BENCH_REGION_TRACE=1 python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 2> $E/trace_1.err | tail -1 > $E/b_1.json
grep region-trace $E/trace_1.err | head -2 | cut -c1-400
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $E/bench_$i.json
python - <<PY
import json
b = json.load(open("$E/bench_$i.json"))
print(json.dumps(b["summary"]))
for c in b.get("other_configs", []):
    print(c.get("config"), c.get("ms_per_step"), c.get("ms_per_step_amortised"), c.get("raster_ms_steady_median"), c.get("host_ms_to_queue_one_iteration"), c.get("host_ms_in_step_call_incl_gpu_wait"), c.get("pool_bytes_reserved"), (c.get("allocator") or {}).get("reserved_bytes.all.peak"), c.get("device_allocations_in_region"), c.get("other_run"))
PY
done
