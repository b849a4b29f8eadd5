#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_primitives.py -q --tb=short -m gpu 2>&1 | grep -v amdgpu.ids | tail -4
timeout 900 python -m pytest tests/test_gpu_parity.py -q --tb=short -m gpu -k "not full_size" 2>&1 | grep -v amdgpu.ids | tail -8
run() { timeout 600 python bench.py --no-cpu-baseline "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['stages_ms'])"; }
echo "== bench"; run; run --loss scalar
