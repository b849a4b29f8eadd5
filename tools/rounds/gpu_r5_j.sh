#!/bin/bash
R=$GRAFT_REPO_ROOT; E=$R/gpurun_out/r5_j; mkdir -p $E; cd $R
timeout 300 python tools/densify_gc_trace.py 1000000 2>&1 | grep -v amdgpu.ids | tee $E/gc_trace.txt | cut -c1-400
