#!/bin/bash
# round 4, call C: scalar walk with the record prefetch (shipped) against the no-prefetch variant and the LDS kernel;
# SQ counters of the three on a short bench; the multi-view SH rebuild at V = 8
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/r4c
mkdir -p $E
cd $R
echo "== parity (quick)"
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_parity.py tests/test_gpu_multiview.py -x -q --tb=short -m gpu 2>&1 | grep -v amdgpu.ids | tail -8 | tee $E/pytest_parity.log
run() {  # tag, extra bench args
  python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other-configs $2 2>/dev/null | tail -1 > $E/bench_$1.json
  python - <<PY
import json
b = json.load(open("$E/bench_$1.json"))
print("$1", "value", b["value"], "ms", b["ms_per_step"], "exact", b.get("value_exact"), b.get("ms_per_step_exact"), {k: (b.get("parity_mode") or {}).get(k) for k in ("blend_fwd_ms", "blend_bwd_ms", "gauss_bwd_ms")}, b["roofline"]["stages_ms"])
PY
}
export SGR_BINDING=ctypes
for rep in 1 2; do
  run sw_$rep ""
  SGR_LIB=$R/street_gaussians_amd/variants/libsgr_hip_swnopf.so run swnopf_$rep ""
  SGR_NO_SW=1 run lds_$rep ""
done
echo "== SQ counters (short bench), scalar walk vs LDS kernel"
cd /tmp && export TMPDIR=/tmp
for tag in sw lds; do
  if [ $tag = lds ]; then export SGR_NO_SW=1; else unset SGR_NO_SW; fi
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $E/pmc_$tag -o sq -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $E/pmc_$tag.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $E/pmc2_$tag -o sq -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $E/pmc2_$tag.log 2>&1
done
unset SGR_NO_SW
cd $R
python - <<'PY'
import csv, glob, collections, os
E = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r4c")
for d in sorted(glob.glob(E + "/pmc*_*")):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "blend_bwd" in k or "row_sum" in k:
                acc[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        print(os.path.basename(d), k, {n: round(sum(v) / len(v)) for n, v in c.items()})
PY
echo "== SH rebuild from V views (1 M Gaussians)"
python tools/bench_mv.py 2>&1 | grep -v amdgpu.ids | tail -3 | tee $E/mv.json
rm -rf $E/pmc_*/*/*kernel_trace.csv $E/pmc2_*/*/*kernel_trace.csv
