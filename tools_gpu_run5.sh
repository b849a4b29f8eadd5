#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_primitives.py -q --tb=short -m gpu -k wave_sum 2>&1 | grep -v amdgpu.ids | tail -5
timeout 900 python -m pytest tests/test_gpu_parity.py -q --tb=short -m gpu -k "backward or culling or reference or golden or autograd or precomputed" 2>&1 | grep -v amdgpu.ids | tail -15
echo "== bench (det)"; timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['stages_ms'])"
echo "== bench (no det)"; SGR_NO_DET=1 timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['stages_ms'])"
