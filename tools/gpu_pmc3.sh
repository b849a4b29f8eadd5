#!/bin/bash
# Round-3 counter passes (each `--pmc` group in its own run, with --kernel-trace only): SQ issue / wait / VALU-busy /
# LDS counters + the GRBM clock for the blend kernels at 1 M Gaussians, FETCH_SIZE / WRITE_SIZE at 1 M and 5 M for the
# streaming kernels.  usage: bash tools/gpu_pmc3.sh <tag> [sizes, default "1000000 5000000"]
R=$GRAFT_REPO_ROOT
TAG=${1:-r3}
SIZES=${2:-"1000000 5000000"}
cd /tmp && export TMPDIR=/tmp
for P in $SIZES; do
  D=$R/gpurun_out/pmc3_${TAG}_$P
  mkdir -p $D
  export SGR_BENCH_P=$P
  i=0
  if [ "$P" = "1000000" ]; then
    SETS=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
          "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE GRBM_COUNT"
          "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"
          "FETCH_SIZE" "WRITE_SIZE")
  else
    SETS=("FETCH_SIZE" "WRITE_SIZE")
  fi
  for set in "${SETS[@]}"; do
    i=$((i+1))
    timeout 420 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $D/g$i -o p -- python $R/profiles/pmc_workload.py > $D/g$i.log 2>&1
    echo "P=$P set $i rc=$? $(tail -1 $D/g$i.log | cut -c1-100)"
  done
  rm -f $D/*/*_kernel_trace.csv $D/*/*/*_kernel_trace.csv
  python $R/tools/pmc_table.py $D $D/table.json > /dev/null 2>&1
  ls $D
done
