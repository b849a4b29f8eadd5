#!/bin/bash
mkdir -p gpurun_out
echo "== full gpu suite"; timeout 1500 python -m pytest tests -q --tb=short -m gpu 2>&1 | grep -v amdgpu.ids | tail -40 | tee gpurun_out/t_gpu_all.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -5
echo "== bench"; timeout 900 python bench.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee gpurun_out/bench_r1_a.json
cd /tmp && export TMPDIR=/tmp
echo "== rocprof stats"; timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_stats.log 2>&1
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof_stats | head -20
