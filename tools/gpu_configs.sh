#!/bin/bash
# bench.py over BASELINE.json's other single-GPU configurations (parity-test sizes, reported for reference)
run() { echo "== $*"; timeout 900 python bench.py --no-cpu-baseline "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['num_rendered_R'], d['roofline']['stages_ms'])"; }
run --gaussians 500000
run --gaussians 2000000 --semantics 19
run --gaussians 5000000 --steps 10 --warmup 3
run --gaussians 1000000 --width 3840 --height 2160 --steps 10 --warmup 3
