#!/bin/bash
# Copies what tools/gpu_evidence.sh <tag> left under gpurun_out/ev_<tag>/ into profiles/ (tracked): run here after gpurun.
# usage: bash tools/collect_evidence.sh <tag> [round-dir, default r6]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
E=$R/gpurun_out/ev_$1
D=$R/profiles/${2:-r6}
N=${2:-r6}
mkdir -p $D
cp $E/pmc_blend_bwd.json $R/profiles/pmc_blend_bwd.json
cp $E/pmc_blend_bwd.json $D/pmc_blend_bwd.json
cp $E/bench.json $D/bench_$N.json
cp $E/stats/*kernel_stats.csv $D/bench_kernel_stats_$N.csv 2>/dev/null || cp $E/stats/*/*kernel_stats.csv $D/bench_kernel_stats_$N.csv
for c in 5m 2m; do f=$(ls $E/stats$c/*kernel_stats.csv $E/stats$c/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $D/bench_kernel_stats_${N}_$c.csv; done
cp $E/pytest_gpu.log $D/pytest_gpu_$N.log
cp $E/pmc_table_1M.json $E/pmc_table_5M.json $E/fullsize_parity.json $E/parity_measured.jsonl $D/
[ -f $E/threeway_fullsize.json ] && cp $E/threeway_fullsize.json $D/ || cp $R/gpurun_out/threeway_fullsize.json $D/
for t in scene densify iteration loss binding; do [ -f $E/$t.json ] && cp $E/$t.json $D/${t}_$N.json; done
python - <<PY
import json, sys
sys.path.insert(0, "$R")
from street_gaussians_amd import build
have = json.load(open("$R/profiles/pmc_blend_bwd.json")).get("source_sha16")
print("source hash", build.source_sha16(), "traffic file", have, "OK" if have == build.source_sha16() else "MISMATCH")
PY
