#!/bin/bash
# A/B of library variants (tools/build_variant.py): bench stage times with the shipped build and with each variant
# usage: bash tools/gpu_ab_lib.sh "<bench args>" <variant names...>
R=$GRAFT_REPO_ROOT
ARGS=$1; shift
run() {
  python $R/bench.py --no-cpu-baseline --no-other-configs $ARGS 2>/dev/null | tail -1 | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('$1', b['ms_per_step'], (b.get('sustained') or {}).get('ms_per_step'), b['roofline']['stages_ms'])"
}
for rep in 1 2; do
  SGR_BINDING=ctypes run shipped
  for v in "$@"; do SGR_BINDING=ctypes SGR_LIB=$R/street_gaussians_amd/variants/libsgr_hip_$v.so run $v; done
done
