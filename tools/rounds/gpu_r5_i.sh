#!/bin/bash
R=$GRAFT_REPO_ROOT; E=$R/gpurun_out/r5_i; mkdir -p $E; cd $R
timeout 600 python tools/densify_mem_trace.py 5000000 2>&1 | grep -v amdgpu.ids | tee $E/mem_trace.txt
