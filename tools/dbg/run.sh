#!/bin/bash
timeout 900 python bench.py 2>&1 | grep -v amdgpu.ids | tail -3 > gpurun_out/bench_r2_try.json; tail -c 6000 gpurun_out/bench_r2_try.json
timeout 300 python -m pytest tests/test_gpu_parity.py -q --tb=short -m gpu -x -k "backward_matches or edge or random" 2>&1 | grep -v amdgpu.ids | tail -8
