#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/trace -o t -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $R/gpurun_out/trace.log 2>&1
ls -la $R/gpurun_out/trace
