#!/bin/bash
run() { timeout 600 python bench.py --no-cpu-baseline --no-other-configs --steps 200 "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['stages_ms'])"; }
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_parity.py -q --tb=short -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -4
echo "== local shuffle"; run
run --gaussians 5000000
