#!/bin/bash
# TEST INFRASTRUCTURE.  Secondary oracle ("oracle/_ref"): compiles the reference's own, unmodified
# CUDA sources for gfx950 from where they lie under /root/reference and links them with
# oracle/ref_wrap.hip into oracle/_ref/libref_rasterizer.so.  Nothing from /root/reference is
# copied into the repo: the only transformation is nvcc's `<< <` / `>> >` launch-token spelling
# (clang needs `<<<` / `>>>`), applied by sed into a mktemp dir that is removed afterwards;
# CUDA runtime / CUB / cooperative-groups names are mapped to HIP by the headers in oracle/ref_shim/.
# Built with -ffp-contract=off like oracle/sgr_oracle.c so integer outputs are reproducible.
# Only runs where /root/reference exists (the authoring container); the .so travels to the GPU box.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${SGR_REFERENCE_ROOT:-/root/reference}"
DGR="$REF/submodules/diff-gaussian-rasterization"
KNN="$REF/submodules/simple-knn"
if [ ! -d "$DGR/cuda_rasterizer" ]; then echo "reference not present at $REF; skipping oracle/_ref"; exit 0; fi
OUT="$HERE/_ref"
mkdir -p "$OUT"
if [ -f "$OUT/libref_rasterizer.so" ] && [ "$OUT/libref_rasterizer.so" -nt "$HERE/ref_wrap.hip" ] && [ "${1:-}" != "-f" ]; then
  echo "oracle/_ref up to date"; exit 0; fi
TMP="$(mktemp -d /tmp/sgr_refbuild.XXXXXX)"
trap 'rm -rf "$TMP"' EXIT
FLAGS="--offload-arch=gfx950 -O3 -fPIC -ffp-contract=off -w -I$HERE/ref_shim -I$DGR/third_party/glm -I$DGR/cuda_rasterizer -I$DGR -I$KNN"
for f in rasterizer_impl forward backward; do
  sed -e 's/<< *</<<</g' -e 's/>> *>/>>>/g' "$DGR/cuda_rasterizer/$f.cu" > "$TMP/$f.hip"
  hipcc $FLAGS -c "$TMP/$f.hip" -o "$TMP/$f.o" &
done
sed -e 's/<< *</<<</g' -e 's/>> *>/>>>/g' "$KNN/simple_knn.cu" > "$TMP/simple_knn.hip"
hipcc $FLAGS -c "$TMP/simple_knn.hip" -o "$TMP/simple_knn.o" &
hipcc $FLAGS -c "$HERE/ref_wrap.hip" -o "$TMP/ref_wrap.o" &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libref_rasterizer.so" "$TMP"/*.o
echo "built $OUT/libref_rasterizer.so"
