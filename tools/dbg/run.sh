#!/bin/bash
run() { timeout 600 python bench.py --no-cpu-baseline --no-other-configs --steps 100 "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['stages_ms'])"; }
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q --tb=short -m gpu -x 2>&1 | grep -v amdgpu.ids | grep -v "^{" | tail -8
echo "== moments"; run
run --gaussians 2000000 --semantics 19
