#!/bin/bash
# stage times at several semantic channel counts, shipped build against a variant library
# usage: bash tools/gpu_ab_sem.sh <variant> "<S values>" [bench args]
R=$GRAFT_REPO_ROOT
V=${1:-fwdold}; SS=${2:-"3 8 12 22"}; shift; shift
for S in $SS; do
  for v in shipped $V; do
    if [ $v = shipped ]; then L=""; else L="SGR_LIB=$R/street_gaussians_amd/variants/libsgr_hip_$v.so"; fi
    env SGR_BINDING=ctypes $L python $R/bench.py --no-cpu-baseline --no-other-configs --steps 100 --semantics $S "$@" 2>/dev/null | tail -1 | python -c "import sys,json; b=json.loads(sys.stdin.read()); st=b['roofline']['stages_ms']; print('S=$S', '$v', b['ms_per_step'], 'fwd', st['blend_fwd'], 'bwd', st['blend_bwd'], 'gauss_bwd', st['gauss_bwd'])"
  done
done
