#!/bin/bash
# round 5, call M: the pruned library (A/B designs out of the shipped build) through the whole -m gpu suite; the same designs'
# tests against a -DSGR_WITH_VARIANTS=1 build; smoke; the default bench line
R=$GRAFT_REPO_ROOT; E=$R/gpurun_out/r5_m; mkdir -p $E; cd $R
rm -f gpurun_out/parity_measured.jsonl gpurun_out/threeway_fullsize.json gpurun_out/fullsize_parity.json
timeout 2400 python -m pytest tests -q --tb=short -m gpu 2>&1 | grep -v amdgpu.ids | grep -v "^{" | tail -30 | tee $E/pytest_gpu.log
cp gpurun_out/parity_measured.jsonl gpurun_out/threeway_fullsize.json gpurun_out/fullsize_parity.json $E/ 2>/dev/null
SGR_BINDING=ctypes SGR_LIB=$R/street_gaussians_amd/variants/libsgr_hip_ab.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_primitives.py -q --tb=short -m gpu -k "culling or scalar_walk or sort_pairs" 2>&1 | grep -v amdgpu.ids | tail -5 | tee $E/pytest_variants.log
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $E/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $E/bench.json
python - <<PY
import json
b = json.load(open("$E/bench.json"))
print(json.dumps(b["summary"]))
for c in b.get("other_configs", []):
    print(c.get("config"), c.get("ms_per_step"), c.get("ms_per_step_amortised"), c.get("host_ms_to_queue_one_iteration"), (c.get("allocator") or {}).get("reserved_bytes.all.peak"), c.get("device_allocations_in_region"), c.get("other_run"))
PY
