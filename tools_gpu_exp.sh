#!/bin/bash
run() { timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'fwd', d['roofline']['stages_ms']['blend_fwd'], 'bwd', d['roofline']['stages_ms']['blend_bwd'])"; }
echo "== baseline"; run
echo "== NOREDUCE"; SGR_EXTRA_FLAGS="-DSGR_EXP_NOREDUCE" python -m street_gaussians_amd.build -f > /dev/null 2>&1; run
echo "== NOHIT"; SGR_EXTRA_FLAGS="-DSGR_EXP_NOHIT" python -m street_gaussians_amd.build -f > /dev/null 2>&1; run
echo "== NOCULL baseline build"; python -m street_gaussians_amd.build -f > /dev/null 2>&1; SGR_NO_CULL=1 run
