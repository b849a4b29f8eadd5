#!/bin/bash
# kernel + memory-copy trace of a short bench run (gap analysis between launches); extra bench args pass through
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/trace -o t -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/trace.log 2>&1
tail -2 $R/gpurun_out/trace.log | cut -c1-200
ls -la $R/gpurun_out/trace
