#!/bin/bash
# A/B of run-time switches: bench stage times with the default and with each "VAR=value" setting given
# usage: bash tools/gpu_ab_env.sh "<bench args>" VAR=1 [VAR2=1 ...]
R=$GRAFT_REPO_ROOT
ARGS=$1; shift
run() {
  python $R/bench.py --no-cpu-baseline --no-other-configs $ARGS 2>/dev/null | tail -1 | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('$1', b['ms_per_step'], (b.get('sustained') or {}).get('ms_per_step'), b['roofline']['stages_ms'])"
}
for rep in 1 2; do
  run default
  for v in "$@"; do env $v bash -c "$(declare -f run); R=$R; ARGS='$ARGS'; run $v"; done
done
