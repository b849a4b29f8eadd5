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
if [ -f "$OUT/libref_rasterizer.so" ] && [ -f "$OUT/libref_rasterizer_fmad.so" ] && [ "$OUT/libref_rasterizer.so" -nt "$HERE/ref_wrap.hip" ] && [ "${1:-}" != "-f" ]; then
  echo "oracle/_ref up to date"; exit 0; fi
TMP="$(mktemp -d /tmp/sgr_refbuild.XXXXXX)"
trap 'rm -rf "$TMP"' EXIT
FLAGS="--offload-arch=gfx950 -O3 -fPIC -ffp-contract=off -w -I$HERE/ref_shim -I$DGR/third_party/glm -I$DGR/cuda_rasterizer -I$DGR -I$KNN"
# Second build of the SAME untouched sources with the compiler's default floating-point contraction (a*b+c may become
# one fused multiply-add wherever the optimiser likes): what the reference's own toolchain does (nvcc's default is
# --fmad=true; its setup.py passes no flag that turns it off).  libref_rasterizer_fmad.so is therefore another VALID
# rounding of the reference algorithm; the distance between the two builds is the reference's own toolchain noise, the
# yardstick tests/test_gpu_fullsize.py holds the HIP path's end-to-end gradient deviation against.
FLAGS_FMAD="${FLAGS/-ffp-contract=off/-ffp-contract=fast}"
mkdir -p "$TMP/fmad"
for f in rasterizer_impl forward backward; do
  sed -e 's/<< *</<<</g' -e 's/>> *>/>>>/g' "$DGR/cuda_rasterizer/$f.cu" > "$TMP/$f.hip"
  hipcc $FLAGS -c "$TMP/$f.hip" -o "$TMP/$f.o" &
  hipcc $FLAGS_FMAD -c "$TMP/$f.hip" -o "$TMP/fmad/$f.o" &
done
sed -e 's/<< *</<<</g' -e 's/>> *>/>>>/g' "$KNN/simple_knn.cu" > "$TMP/simple_knn.hip"
hipcc $FLAGS -c "$TMP/simple_knn.hip" -o "$TMP/simple_knn.o" &
hipcc $FLAGS -c "$HERE/ref_wrap.hip" -o "$TMP/ref_wrap.o" &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libref_rasterizer.so" "$TMP"/*.o
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libref_rasterizer_fmad.so" "$TMP"/fmad/*.o "$TMP/simple_knn.o" "$TMP/ref_wrap.o"
echo "built $OUT/libref_rasterizer.so and $OUT/libref_rasterizer_fmad.so"
