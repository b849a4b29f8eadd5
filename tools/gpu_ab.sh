#!/bin/bash
# A/B of compiler flags on the GPU box: default build first, then a rebuild per argument
run() { timeout 600 python bench.py --no-cpu-baseline "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['stages_ms'])"; }
if [ -z "$SKIP_TESTS" ]; then timeout 900 python -m pytest tests/test_gpu_parity.py -q --tb=short -m gpu -k "not full_size" 2>&1 | grep -v amdgpu.ids | tail -4; fi
echo "== default"; run
for f in "$@"; do
echo "== $f"; SGR_EXTRA_FLAGS="$f" python -m street_gaussians_amd.build -f > /dev/null 2>&1; run
done
