#!/bin/bash
# per-kernel averages (rocprofv3 --kernel-trace --stats) of short bench runs at the given sizes
# usage: bash tools/gpu_kstats.sh <tag> "<sizes>" [ENV=VAL ...]
R=$GRAFT_REPO_ROOT
TAG=${1:-k}
SIZES=${2:-"1000000 5000000"}
shift; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
for P in $SIZES; do
  D=$R/gpurun_out/ks_${TAG}_$P
  mkdir -p $D
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o b -- python $R/bench.py --gaussians $P --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $D/run.log 2>&1
  rm -f $D/*kernel_trace.csv $D/*/*kernel_trace.csv
  echo "== P=$P $@"
  python $R/tools/kstats.py $(ls $D/*kernel_stats.csv $D/*/*kernel_stats.csv 2>/dev/null | head -1) sgr_ | head -30
  tail -1 $D/run.log | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('ms_per_step', b['ms_per_step'], b['roofline']['stages_ms'])"
done
