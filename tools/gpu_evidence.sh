#!/bin/bash
R=$GRAFT_REPO_ROOT
TAG=${1:-v3}
mkdir -p $R/gpurun_out/ev_$TAG
echo "== full gpu suite"; timeout 1500 python -m pytest tests -q --tb=short -m gpu 2>&1 | grep -v amdgpu.ids | tail -6 | tee $R/gpurun_out/ev_$TAG/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
echo "== bench"; timeout 900 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee $R/gpurun_out/ev_$TAG/bench.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ev_$TAG/stats -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/ev_$TAG/stats.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/ev_$TAG/fetch -o fetch -- python $R/profiles/pmc_workload.py > $R/gpurun_out/ev_$TAG/fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/ev_$TAG/write -o write -- python $R/profiles/pmc_workload.py > $R/gpurun_out/ev_$TAG/write.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/gpurun_out/ev_$TAG/sq -o sq -- python $R/profiles/pmc_workload.py > $R/gpurun_out/ev_$TAG/sq.log 2>&1
rm -f $R/gpurun_out/ev_$TAG/*/*_kernel_trace.csv
ls $R/gpurun_out/ev_$TAG/*
