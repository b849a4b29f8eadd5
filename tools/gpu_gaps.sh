#!/bin/bash
# kernel traces of a short bench run with the spinning host wait (default) and with the event wait (SGR_SPIN_US=0):
# tools/gaps.py turns them into "kernel time vs wall time per step" (how much of a step the GPU had nothing queued)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for m in spin event; do
  rm -rf $R/gpurun_out/gaps_$m
  if [ $m = event ]; then export SGR_SPIN_US=0; fi
  timeout 100 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gaps_$m -o t -- python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs > $R/gpurun_out/gaps_$m.log 2>&1
  tail -1 $R/gpurun_out/gaps_$m.log | cut -c1-160
done
