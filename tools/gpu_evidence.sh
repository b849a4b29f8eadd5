#!/bin/bash
# Evidence run on the MI355X box: full `-m gpu` suite, smoke(), the default bench.py line, rocprofv3 kernel stats of a short
# bench run, and the PMC passes (FETCH_SIZE / WRITE_SIZE / SQ, each in its own pass) over profiles/pmc_workload.py.
# Everything lands in gpurun_out/ev_<tag>/; tools/pmc_summary.py turns the PMC passes into profiles/pmc_blend_bwd.json
# (stamped with the kernel-source hash bench.py checks).  Copy what is cited into profiles/<round>/.
R=$GRAFT_REPO_ROOT
TAG=${1:-r6}
E=$R/gpurun_out/ev_$TAG
mkdir -p $E
rm -f $R/gpurun_out/parity_measured.jsonl $R/gpurun_out/fullsize_parity.json $R/gpurun_out/threeway_fullsize.json
echo "== full gpu suite"; timeout 1500 python -m pytest tests -q --tb=short -m gpu 2>&1 | grep -v amdgpu.ids | grep -v "^{" | tail -6 | tee $E/pytest_gpu.log
if [ -f $R/street_gaussians_amd/variants/libsgr_hip_ab.so ]; then
  echo "== the gated A/B designs (library built with -DSGR_WITH_VARIANTS=1)"
  SGR_BINDING=ctypes SGR_LIB=$R/street_gaussians_amd/variants/libsgr_hip_ab.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_primitives.py -q --tb=short -m gpu -k "culling or scalar_walk or sort_pairs" 2>&1 | grep -v amdgpu.ids | tail -3 | tee $E/pytest_variants.log
fi
echo "== soak"; timeout 900 python tools/soak.py 100 2>&1 | grep -v amdgpu.ids | tail -12 | tee $E/soak.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $E/smoke.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $E/stats -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $E/stats.log 2>&1
# per-kernel averages at the other single-GPU configurations too (5 M Gaussians; 2 M + 19 semantic channels)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $E/stats5m -o bench -- python $R/bench.py --gaussians 5000000 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $E/stats5m.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $E/stats2m -o bench -- python $R/bench.py --gaussians 2000000 --semantics 19 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $E/stats2m.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $E/fetch -o fetch -- python $R/profiles/pmc_workload.py > $E/fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $E/write -o write -- python $R/profiles/pmc_workload.py > $E/write.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $E/sq -o sq -- python $R/profiles/pmc_workload.py > $E/sq.log 2>&1
timeout 900 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $E/sq2 -o sq2 -- python $R/profiles/pmc_workload.py > $E/sq2.log 2>&1
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --kernel-trace --output-format csv -d $E/sq3 -o sq3 -- python $R/profiles/pmc_workload.py > $E/sq3.log 2>&1
python $R/tools/pmc_table.py $E $E/pmc_table_1M.json > /dev/null 2>&1
# the streaming kernels at configs[4]'s size: FETCH_SIZE / WRITE_SIZE per kernel at 5 M Gaussians
mkdir -p $E/p5m
SGR_BENCH_P=5000000 timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $E/p5m/fetch -o fetch -- python $R/profiles/pmc_workload.py > $E/p5m/fetch.log 2>&1
SGR_BENCH_P=5000000 timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $E/p5m/write -o write -- python $R/profiles/pmc_workload.py > $E/p5m/write.log 2>&1
rm -f $E/p5m/*/*_kernel_trace.csv
python $R/tools/pmc_table.py $E/p5m $E/pmc_table_5M.json > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $E $E/pmc_blend_bwd.json > /dev/null 2>&1 && cp $E/pmc_blend_bwd.json profiles/pmc_blend_bwd.json
rm -f $E/*/*_kernel_trace.csv
echo "== bench (with the fresh traffic file)"; timeout 1200 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee $E/bench.json | cut -c1-400
echo "== rows next to the path (n1-n4)"
for t in iteration scene loss densify binding; do timeout 600 python tools/bench_$t.py 2>&1 | grep -v amdgpu.ids | tail -1 > $E/$t.json; done
cp $R/gpurun_out/parity_measured.jsonl $R/gpurun_out/fullsize_parity.json $R/gpurun_out/threeway_fullsize.json $E/ 2>/dev/null
ls $E
