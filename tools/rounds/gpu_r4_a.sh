#!/bin/bash
# round 4, call A: the parity mode -- elementary-function self-test, exact legs of the parity suites, three-way at the four
# BASELINE sizes, and the cost of the mode (shipped build vs the library-expf / IEEE-division variant `extrim0`)
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/r4a
mkdir -p $E
cd $R
rm -f gpurun_out/parity_measured.jsonl gpurun_out/threeway_fullsize.json
echo "== exact-math self-test + exact legs"
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_parity.py -q --tb=short -m gpu -k "parity_mode or exact or golden or edge or random" 2>&1 | grep -v amdgpu.ids | tail -25 | tee $E/pytest_exact.log
echo "== three-way at the four BASELINE sizes"
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q --tb=short -m gpu -k threeway 2>&1 | grep -v amdgpu.ids | grep -v "^{" | tail -15 | tee $E/pytest_threeway.log
cp gpurun_out/parity_measured.jsonl gpurun_out/threeway_fullsize.json $E/ 2>/dev/null
run() {
  python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > $E/bench_$1.json
  python - <<PY
import json
b = json.load(open("$E/bench_$1.json"))
print("$1", "value", b["value"], "ms", b["ms_per_step"], "exact", b.get("value_exact"), b.get("ms_per_step_exact"), {k: b["parity_mode"].get(k) for k in ("blend_fwd_ms", "blend_bwd_ms", "gauss_bwd_ms")}, b["roofline"]["stages_ms"])
PY
}
echo "== bench A/B (ctypes binding both)"
for rep in 1 2; do
  SGR_BINDING=ctypes run shipped_$rep
  SGR_BINDING=ctypes SGR_LIB=$R/street_gaussians_amd/variants/libsgr_hip_extrim0.so run extrim0_$rep
done
