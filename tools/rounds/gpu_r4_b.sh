#!/bin/bash
# round 4, call B: the scalar-walk blend backward -- whole GPU suite, then same-box A/B against the LDS-staged kernel
# (SGR_NO_SW=1) at 1 M and 5 M Gaussians, default and parity mode
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/r4b
mkdir -p $E
cd $R
rm -f gpurun_out/parity_measured.jsonl gpurun_out/threeway_fullsize.json gpurun_out/fullsize_parity.json
echo "== primitives + parity first"
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_parity.py -x -q --tb=short -m gpu 2>&1 | grep -v amdgpu.ids | tail -30 | tee $E/pytest_parity.log
echo "== whole suite"
timeout 1500 python -m pytest tests -q --tb=line -m gpu 2>&1 | grep -v amdgpu.ids | grep -v "^{" | tail -30 | tee $E/pytest_gpu.log
cp gpurun_out/parity_measured.jsonl gpurun_out/threeway_fullsize.json gpurun_out/fullsize_parity.json $E/ 2>/dev/null
run() {  # tag, extra bench args
  python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other-configs $2 2>/dev/null | tail -1 > $E/bench_$1.json
  python - <<PY
import json
b = json.load(open("$E/bench_$1.json"))
print("$1", "value", b["value"], "ms", b["ms_per_step"], "exact", b.get("value_exact"), b.get("ms_per_step_exact"), {k: (b.get("parity_mode") or {}).get(k) for k in ("blend_fwd_ms", "blend_bwd_ms", "gauss_bwd_ms")}, b["roofline"]["stages_ms"])
PY
}
echo "== bench A/B"
for rep in 1 2; do
  run sw_1M_$rep ""
  SGR_NO_SW=1 run lds_1M_$rep ""
done
run sw_5M "--gaussians 5000000 --steps 40"
SGR_NO_SW=1 run lds_5M "--gaussians 5000000 --steps 40"
run sw_500k "--gaussians 500000"
SGR_NO_SW=1 run lds_500k "--gaussians 500000"
