#!/bin/bash
# PMC passes over profiles/pmc_workload.py (counters only with --kernel-trace; one group per pass)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_INSTS_VALU_TRANS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc/g$i -o p -- python $R/profiles/pmc_workload.py > $R/gpurun_out/pmc/g$i.log 2>&1
  tail -1 $R/gpurun_out/pmc/g$i.log
done
ls $R/gpurun_out/pmc/*
