#!/bin/bash
# round 2, run A: correctness of the new backward walk (hit record, folded reduction, S>8 determinism), full-size
# oracle parity with measured errors, and A/B of the new pieces.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2a
mkdir -p $O
rm -f $R/gpurun_out/parity_measured.jsonl $R/gpurun_out/fullsize_parity.json
run() { timeout 600 python bench.py --no-cpu-baseline --steps 40 "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['stages_ms'])"; }
echo "== primitives + parity"; timeout 1200 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_parity.py -q --tb=short -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -15 | tee $O/pytest_parity.log
echo "== full size"; timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q --tb=short -m gpu 2>&1 | grep -v amdgpu.ids | tail -15 | tee $O/pytest_fullsize.log
echo "== rest of the gpu suite"; timeout 1200 python -m pytest tests -q --tb=short -m gpu --deselect tests/test_gpu_parity.py --deselect tests/test_gpu_fullsize.py --deselect tests/test_gpu_primitives.py 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/pytest_rest.log
echo "== bench default build"; run | tee $O/ab.log
echo "== no hit record (geometric cull in the backward)"; SGR_NO_HITS=1 run | tee -a $O/ab.log
echo "== -DSGR_FOLD=0"; SGR_EXTRA_FLAGS="-DSGR_FOLD=0" python -m street_gaussians_amd.build -f > /dev/null 2>&1; run | tee -a $O/ab.log
python -m street_gaussians_amd.build -f > /dev/null 2>&1
echo "== configs"; for cfg in "--gaussians 500000" "--gaussians 2000000 --semantics 19" "--gaussians 5000000"; do echo $cfg; run $cfg | tee -a $O/configs.log; done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/stats.log 2>&1
rm -f $O/stats/*/*_kernel_trace.csv $O/stats/*_kernel_trace.csv
ls -R $O | head -30
