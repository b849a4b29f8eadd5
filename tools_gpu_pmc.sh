#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -E "^\s*(SQ_INSTS_VALU|SQ_INSTS_SALU|SQ_INSTS_LDS|SQ_WAVE_CYCLES|SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_ANY|SQ_WAIT_ANY|SQ_WAVES|SQ_BUSY_CYCLES|SQ_LDS_BANK_CONFLICT|SQ_ACTIVE_INST_VALU|SQ_ACTIVE_INST_LDS|SQ_INST_CYCLES_VMEM|SQ_INSTS_VMEM|SQ_WAIT_INST_LDS|SQ_LDS_IDX_ACTIVE|SQ_INSTS_VALU_TRANS|SQ_THREAD_CYCLES_VALU|GRBM_GUI_ACTIVE)\b" | head -30
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc/$tag -o p -- python $R/profiles/pmc_workload.py > $R/gpurun_out/pmc/$tag.log 2>&1
  tail -2 $R/gpurun_out/pmc/$tag.log
done
ls $R/gpurun_out/pmc/*
