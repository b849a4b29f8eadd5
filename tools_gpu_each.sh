#!/bin/bash
# run every parity test in its own process (a GPU fault aborts the interpreter), keep full logs
mkdir -p gpurun_out/each
python -m pytest tests/test_gpu_parity.py -m gpu --collect-only -q 2>/dev/null | grep "::" | grep -v full_size > gpurun_out/each/ids.txt
n=0
while read id; do
  n=$((n+1))
  log=gpurun_out/each/$(printf "%02d" $n).log
  echo "### $id" > $log
  timeout 300 python -m pytest "$id" -q --tb=long -m gpu >> $log 2>&1
  rc=$?
  echo "rc=$rc $id"
  if [ $rc -ne 0 ]; then grep -E "Error|error|assert|fault|Fault|^E " $log | head -12 | cut -c1-400; fi
done < gpurun_out/each/ids.txt
