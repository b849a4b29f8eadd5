#!/bin/bash
# full `-m gpu` suite (summary line only) + the default bench line's headline figures; used while iterating
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -q --tb=short -m gpu -x 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|rror|assert" | tail -8
timeout 600 python bench.py --no-cpu-baseline --no-other-configs "$@" 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/bench_check.json
python -c "
import json; d=json.load(open('gpurun_out/bench_check.json')); print(d['value'], d['ms_per_step'], d['sustained'], d['roofline']['kernel_ms']); print(d['roofline']['stages_ms'])"
