#!/bin/bash
run() { timeout 600 python bench.py --no-cpu-baseline --steps 30 "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['stages_ms'])"; }
echo "== default (wide batch 64)"; python tools/dbg/dbg_s19b.py 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-80
run --gaussians 2000000 --semantics 19
run
for f in "-DSGR_BWD_BATCH_WIDE=128" "-DSGR_BWD_BATCH=64"; do
echo "== $f"; SGR_EXTRA_FLAGS="$f" python -m street_gaussians_amd.build -f > /dev/null 2>&1; python tools/dbg/dbg_s19b.py 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-80
run --gaussians 2000000 --semantics 19
done
