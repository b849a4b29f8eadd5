#!/bin/bash
mkdir -p gpurun_out
for id in "tests/test_gpu_parity.py::test_backward_matches_oracle" "tests/test_gpu_parity.py::test_culling_is_invisible_and_backward_is_deterministic" "tests/test_gpu_parity.py::test_against_reference_kernels"; do
  timeout 600 python -m pytest "$id" -q --tb=short -m gpu 2>&1 | grep -v amdgpu.ids | tail -25
done
echo "== bench (det)"; timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -3
echo "== bench (no det)"; SGR_NO_DET=1 timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -3
echo "== bench (no cull)"; SGR_NO_CULL=1 timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -3
echo "== bench (no dpp)"; SGR_NO_DPP=1 timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -3
