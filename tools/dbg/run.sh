#!/bin/bash
run() { timeout 600 python bench.py --no-cpu-baseline --no-other-configs --steps 200 "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['stages_ms'])"; }
for ipt in 8 16 12; do
echo "== IPT $ipt"; SGR_EXTRA_FLAGS="-DSGR_SORT_IPT=$ipt" python -m street_gaussians_amd.build -f > /dev/null 2>&1
timeout 300 python -m pytest tests/test_gpu_primitives.py -q -m gpu 2>&1 | tail -1
run; run --gaussians 5000000
done
