#!/bin/bash
# The library's own exchange work at ONE rank: bench.py with a forced one-rank RCCL group against the plain run, same box,
# interleaved twice (profiles/r5/bench_forced_one_rank_group.json: 1.627 vs 1.407 ms).  gpurun -- 'bash tools/gpu_forced_rank.sh <tag> [mode]'
R=$GRAFT_REPO_ROOT; TAG=${1:-forced}; MODE=${2:-fast}; E=$R/gpurun_out/$TAG; mkdir -p $E; cd $R
one() { python bench.py --no-cpu-baseline --no-other-configs --mode $MODE --steps 300 "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
b = json.loads(sys.stdin.read())
print(json.dumps({'label': '$LBL', 'mode': b['mode'], 'ms_per_step': b['ms_per_step'], 'dist_ranks': b['dist_ranks'], 'reduce': '$RED',
                  'overlap_ms': (b.get('exchange_overlap') or {}).get('ms_per_step'), 'direct': b.get('exchange_direct_bucket_writes'),
                  'stages': b['roofline']['stages_ms']}))"; }
for rep in 1 2; do
  LBL=plain RED=- one | tee -a $E/forced.jsonl
  LBL=forced RED=factored SGR_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 WORLD_SIZE=1 one --reduce factored | tee -a $E/forced.jsonl
  LBL=forced RED=bucket SGR_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29542 RANK=0 WORLD_SIZE=1 one --reduce bucket | tee -a $E/forced.jsonl
done
