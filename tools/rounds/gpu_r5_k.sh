#!/bin/bash
R=$GRAFT_REPO_ROOT; E=$R/gpurun_out/r5_k; mkdir -p $E; cd $R
timeout 300 python tools/densify_gc_trace.py 1000000 2>&1 | grep -v amdgpu.ids | head -8 | cut -c1-300
timeout 600 python tools/densify_mem_trace.py 5000000 2>&1 | grep -v amdgpu.ids | tee $E/mem_trace.txt | tail -12
timeout 600 python -m pytest tests/test_gpu_densify_loop.py tests/test_gpu_multiview.py tests/test_gpu_callsite.py -q -x -m gpu 2>&1 | grep -v amdgpu.ids | tail -3
