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
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w -x hip -I"$TMP" "$HERE/ref_driver.cpp" \
  -o "$HERE/_ref/ref_driver" -L/opt/rocm/lib -lhipfft -Wl,-rpath,/opt/rocm/lib
echo "built $HERE/_ref/ref_driver"
