#!/bin/bash
# build_ref.sh — builds oracle/_ref/ref_driver: the reference's OWN kernels for gfx950.
# TEST INFRASTRUCTURE ONLY (golden-vector generation on the GPU box; see ref_driver.cpp).
#
# The reference is CUDA.  AMD's hipify-perl (shipped in this image, /opt/rocm/bin) renames the
# cuda*/cufft* API identifiers to hip*/hipfft*; the only other edit is removing the blanks the
# reference has inside its launch brackets (`<< <grid, threads >> >`, 13 sites), which nvcc
# tolerates and clang does not.  No stand-in headers, libraries or tools are written: the
# translated files include <hip/hip_runtime.h> and <hipfft/hipfft.h> from the image and link
# against the image's libhipfft.  The translated text lives only in a temporary directory.
# Output: oracle/_ref/ref_driver (git-ignored, travels to the GPU box like our own .so).
#
# usage: build_ref.sh [NXxNYxNZ ...]
#   The reference's grid is compile-time (LBM.h:32-35) and its box lengths are literals that
#   "need to change according to NX and LX" (LBM.h:40-45).  For every extra grid named on the
#   command line the TEMPORARY copy of LBM.h gets NX, NY, NZ and Lx = NX dx, Ly = NY dy,
#   Lz = (NZ-1) dz rewritten by sed ("Grid and physics are changed by sed on the
#   copied LBM.h") and a second binary oracle/_ref/ref_driver_NXxNYxNZ is built.  NX must be a
#   multiple of nThreads = 10 (LBM.h:29).  Nothing under /root/reference is touched.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${EKPNP_REFERENCE:-/root/reference}"
if [ ! -f "$REF/LBM.cu" ]; then
  echo "build_ref.sh: $REF not present (GPU box?) - keeping prebuilt oracle/_ref if any"
  exit 0
fi
TMP="$(mktemp -d)"
trap 'rm -rf "$TMP"' EXIT
for f in LBM.h LBM.cu poisson.cu; do
  /opt/rocm/bin/hipify-perl "$REF/$f" 2>/dev/null \
    | sed -E 's/<<[[:space:]]*</<<</g; s/>>[[:space:]]*>/>>>/g' > "$TMP/$f"
done
mkdir -p "$HERE/_ref"
build() {  # $1 = include dir, $2 = output
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w -x hip -I"$1" "$HERE/ref_driver.cpp" \
    -o "$2" -L/opt/rocm/lib -lhipfft -Wl,-rpath,/opt/rocm/lib
  echo "built $2"
}
build "$TMP" "$HERE/_ref/ref_driver"
for grid in "$@"; do
  IFS=x read -r gx gy gz <<< "$grid"
  if [ $((gx % 10)) -ne 0 ]; then echo "build_ref.sh: NX=$gx is not a multiple of nThreads=10 (LBM.h:29)"; exit 1; fi
  mkdir -p "$TMP/$grid"
  cp "$TMP/LBM.cu" "$TMP/poisson.cu" "$TMP/$grid/"
  # dx = dy = dz = 1.0e-6/100.0 (LBM.h:43-45); Lx = NX*dx etc. as shortest round-trip decimal literals
  # (the same doubles ekpnp_default_params / oracle_default_params compute)
  lx=$(python3 -c "print(repr($gx * (1.0e-6 / 100.0)))")
  ly=$(python3 -c "print(repr($gy * (1.0e-6 / 100.0)))")
  lz=$(python3 -c "print(repr(($gz - 1) * (1.0e-6 / 100.0)))")
  sed -E \
    -e "s/^(const unsigned int NX = )[0-9]+;/\1$gx;/" \
    -e "s/^(const unsigned int NY = )[0-9]+;/\1$gy;/" \
    -e "s/^(const unsigned int NZ = )[0-9]+;/\1$gz;/" \
    -e "s/^(__constant__ double Lx = )[^;]+;/\1$lx;/" \
    -e "s/^(__constant__ double Ly = )[^;]+;/\1$ly;/" \
    -e "s/^(__constant__ double Lz = )[^;]+;/\1$lz;/" \
    "$TMP/LBM.h" > "$TMP/$grid/LBM.h"
  grep -q "NX = $gx;" "$TMP/$grid/LBM.h" && grep -q "NZ = $gz;" "$TMP/$grid/LBM.h" || { echo "build_ref.sh: grid rewrite failed"; exit 1; }
  build "$TMP/$grid" "$HERE/_ref/ref_driver_$grid"
done
