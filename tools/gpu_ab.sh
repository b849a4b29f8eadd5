#!/bin/bash
# A/B of compiler flags on the GPU box: default build first, then a rebuild with $1
run() { timeout 600 python bench.py --no-cpu-baseline "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['stages_ms'])"; }
timeout 900 python -m pytest tests/test_gpu_parity.py -q --tb=short -m gpu -k "not full_size" 2>&1 | grep -v amdgpu.ids | tail -4
echo "== default"; run; run
echo "== $1"; SGR_EXTRA_FLAGS="$1" python -m street_gaussians_amd.build -f > /dev/null 2>&1; run; run
