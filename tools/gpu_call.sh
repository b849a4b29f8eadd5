#!/bin/bash
# One GPU-box call, by recipe.  Replaces the per-round one-off scripts (tools/gpu_r2..r5_*.sh; each is in the history at the
# commit profiles/README.md names next to the record it produced).  Run from the repo root through gpurun:
#
#   gpurun --timeout 3000 -- 'bash tools/gpu_call.sh <tag> <recipe> [args] [-- <recipe> [args]] ...'
#
# Everything a recipe writes lands in gpurun_out/<tag>/ (copy what is cited into profiles/<round>/).  Recipes:
#
#   check [bench args]            whole `-m gpu` suite, the gated A/B designs' tests against variants/libsgr_hip_ab.so when it was
#                                 built, smoke(), one full default bench line (+ its `summary`)
#   tests <lib|shipped> <-k expr> pytest -m gpu -k <expr> against a library of street_gaussians_amd/variants/ (ctypes binding)
#   ab-lib "<bench args>" <names> stage times of the shipped build and of each variants/libsgr_hip_<name>.so, twice, interleaved
#   ab-env "<bench args>" VAR=1.. the same for run-time switches the library reads from the environment
#   soak <reps> [lib]             tools/soak.py (bit-identity of repeated forward / backward runs)
#   ubench                        tools/ubench/valu_rates2 (instruction costs in measured cycles)
#   kstats "<sizes>" [ENV=VAL..]  rocprofv3 --kernel-trace --stats per-kernel averages of short bench runs at those sizes
#   py <script> [args]            any python tool of this directory (densify_mem_trace.py, bench_mv.py, ...)
R=$GRAFT_REPO_ROOT; TAG=${1:-call}; shift
E=$R/gpurun_out/$TAG; mkdir -p $E; cd $R
V=$R/street_gaussians_amd/variants
clean() { grep -v amdgpu.ids; }
lib_env() { if [ "$1" = shipped ]; then echo "SGR_BINDING=ctypes"; else echo "SGR_BINDING=ctypes SGR_LIB=$V/libsgr_hip_$1.so"; fi; }
stage_line() {  # one bench run -> one JSON line of the figures the A/Bs compare (SGR_AB_MODE: bench --mode, default strict)
  python $R/bench.py --no-cpu-baseline --no-other-configs --mode ${SGR_AB_MODE:-strict} $2 2>/dev/null | tail -1 | python -c "
import sys, json
b = json.loads(sys.stdin.read()); st = b['roofline']['stages_ms']; m = b.get('modes') or {}
chain = sum(st[k] or 0 for k in ('scan', 'duplicate', 'sort', 'tile_ranges'))
print(json.dumps({'variant': '$1', 'mode': b['mode'], 'ms': b['ms_per_step'], 'strict_ms': b.get('ms_per_step_strict'),
                  'exact_ms': b.get('ms_per_step_exact'), 'fast_ms': b.get('ms_per_step_fast'), 'binning_chain_ms': round(chain, 4),
                  'stages': st, 'stages_fast': (m.get('fast') or {}).get('stages_ms'), 'kernel_ms': b['roofline']['kernel_ms']}))"
}
recipe_check() {
  rm -f gpurun_out/parity_measured.jsonl gpurun_out/threeway_fullsize.json gpurun_out/fullsize_parity.json
  timeout 2400 python -m pytest tests -q --tb=short -m gpu 2>&1 | clean | grep -v "^{" | tail -30 | tee $E/pytest_gpu.log
  cp gpurun_out/parity_measured.jsonl gpurun_out/threeway_fullsize.json gpurun_out/fullsize_parity.json $E/ 2>/dev/null
  if [ -f $V/libsgr_hip_ab.so ]; then
    env $(lib_env ab) timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_primitives.py -q --tb=short -m gpu \
      -k "culling or scalar_walk or sort_pairs" 2>&1 | clean | tail -5 | tee $E/pytest_variants.log
  fi
  timeout 600 python __graft_entry__.py smoke 2>&1 | clean | tail -2 | tee $E/smoke.log
  timeout 1200 python bench.py "$@" 2>/dev/null | tail -1 > $E/bench.json
  python - <<PY
import json
b = json.load(open("$E/bench.json"))
print(json.dumps(b["summary"]))
for c in b.get("other_configs", []):
    print(c.get("config"), c.get("ms_per_step"), c.get("ms_per_step_amortised"), c.get("host_ms_to_queue_one_iteration"),
          (c.get("allocator") or {}).get("reserved_bytes.all.peak"), c.get("device_allocations_in_region"))
PY
}
recipe_tests() { env $(lib_env $1) timeout 1200 python -m pytest tests -q --tb=short -m gpu -k "$2" 2>&1 | clean | tail -8 | tee $E/pytest_$1.log; }
recipe_ab_lib() {
  ARGS=$1; shift
  for rep in 1 2; do
    env $(lib_env shipped) bash -c "$(declare -f stage_line); R=$R; stage_line shipped '$ARGS'" | tee -a $E/ab.jsonl
    for v in "$@"; do env $(lib_env $v) bash -c "$(declare -f stage_line); R=$R; stage_line $v '$ARGS'" | tee -a $E/ab.jsonl; done
  done
}
recipe_ab_env() {
  ARGS=$1; shift
  for rep in 1 2; do
    stage_line default "$ARGS" | tee -a $E/ab_env.jsonl
    for v in "$@"; do env $v bash -c "$(declare -f stage_line); R=$R; stage_line $v '$ARGS'" | tee -a $E/ab_env.jsonl; done
  done
}
recipe_soak() { env $(lib_env ${2:-shipped}) timeout 1500 python tools/soak.py $1 2>&1 | clean | tail -12 | tee $E/soak_${2:-shipped}.log; }
recipe_ubench() { timeout 600 tools/ubench/valu_rates2 > $E/valu_rates2.jsonl 2> $E/ubench.err; tail -3 $E/valu_rates2.jsonl; }
recipe_kstats() {
  SIZES=$1; shift
  ( for kv in "$@"; do export "$kv"; done; cd /tmp && export TMPDIR=/tmp
    for P in $SIZES; do
      D=$E/ks_$P; mkdir -p $D
      timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o b -- python $R/bench.py --gaussians $P --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $D/run.log 2>&1
      rm -f $D/*kernel_trace.csv $D/*/*kernel_trace.csv
      echo "== P=$P $@"; python $R/tools/kstats.py $(ls $D/*kernel_stats.csv $D/*/*kernel_stats.csv 2>/dev/null | head -1) sgr_ | head -30
    done )
}
recipe_py() { S=$1; shift; timeout 1200 python tools/$S "$@" 2>&1 | clean | tee $E/$(basename $S .py).txt | tail -40; }

args=()
run_one() { [ ${#args[@]} -eq 0 ] && return; name=${args[0]//-/_}; echo "== ${args[*]}"; recipe_$name "${args[@]:1}"; args=(); }
for a in "$@"; do if [ "$a" = "--" ]; then run_one; else args+=("$a"); fi; done
run_one
