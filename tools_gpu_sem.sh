#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py -q --tb=short -m gpu -k "sem or tiny or mid_20k or smoke or precomputed or golden or full_size" 2>&1 | grep -v amdgpu.ids | tail -8
run() { timeout 900 python bench.py --no-cpu-baseline "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'R', d['config']['num_rendered_R'], d['roofline']['stages_ms'])"; }
echo "== 2M S=19"; run --gaussians 2000000 --semantics 19 --steps 5 --warmup 2
echo "== 1M S=3"; run --semantics 3 --steps 10 --warmup 2
echo "== 1M S=0"; run
