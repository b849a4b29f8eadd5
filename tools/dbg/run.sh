#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in v2 v1; do mkdir -p $R/gpurun_out/pmc_$v
  if [ $v = v1 ]; then export SGR_NO_V2=1; else unset SGR_NO_V2; fi
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_INSTS_VALU_TRANS"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$v/g$i -o p -- python $R/profiles/pmc_workload.py > $R/gpurun_out/pmc_$v/g$i.log 2>&1
  done
done
cd $R
python - <<'PY'
import csv,glob,collections
for v in ("v2","v1"):
    acc=collections.defaultdict(lambda:[0.0,0])
    for f in glob.glob(f"gpurun_out/pmc_{v}/g*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "blend_bwd" in row["Kernel_Name"]:
                a=acc[row["Counter_Name"]]; a[0]+=float(row["Counter_Value"]); a[1]+=1
    print(v, {k: round(a[0]/a[1]) for k,a in sorted(acc.items())})
PY
