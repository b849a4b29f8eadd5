#!/bin/bash
# Round-3 iteration loop on the MI355X box: the new tests first (fail fast), then the whole `-m gpu` suite, then the default
# bench line.  usage: bash tools/gpu_r3_check.sh <tag> [pytest -k filter for the first stage]
R=$GRAFT_REPO_ROOT
TAG=${1:-r3}
E=$R/gpurun_out/chk_$TAG
mkdir -p $E
rm -f $R/gpurun_out/parity_measured.jsonl $R/gpurun_out/fullsize_parity.json $R/gpurun_out/threeway_fullsize.json
echo "== new tests"; timeout 900 python -m pytest tests/test_gpu_densify_loop.py -x -q --tb=short -m gpu 2>&1 | grep -v amdgpu.ids | tail -15 | tee $E/new_tests.log
echo "== full gpu suite"; timeout 1500 python -m pytest tests -q --tb=short -m gpu 2>&1 | grep -v amdgpu.ids | grep -v "^{" | tail -15 | tee $E/pytest_gpu.log
echo "== bench"; timeout 900 python bench.py 2>$E/bench.err | grep -v amdgpu.ids | tail -1 > $E/bench.json; cut -c1-600 $E/bench.json; tail -3 $E/bench.err
cp $R/gpurun_out/parity_measured.jsonl $R/gpurun_out/fullsize_parity.json $R/gpurun_out/threeway_fullsize.json $E/ 2>/dev/null
ls $E
