#!/bin/bash
mkdir -p gpurun_out/prof
echo "== huge_splats tests"; timeout 600 python -m pytest tests/test_gpu_parity.py -q --tb=short -m gpu -k huge_splats 2>&1 | grep -v amdgpu.ids | tail -8
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
echo "== rocprof stats"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/stats -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof/stats.log 2>&1
echo "== pmc fetch"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof/fetch -o fetch -- python $R/profiles/pmc_workload.py > $R/gpurun_out/prof/fetch.log 2>&1
echo "== pmc write"
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof/write -o write -- python $R/profiles/pmc_workload.py > $R/gpurun_out/prof/write.log 2>&1
find $R/gpurun_out/prof -type f | head -30; du -sh $R/gpurun_out/prof
