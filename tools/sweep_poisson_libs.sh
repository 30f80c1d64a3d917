#!/bin/bash
# sweep_poisson_libs.sh "v1 v2 ..." [grid] — per-kernel Poisson times (rocprofv3) of differently built
# libekpnp_<v>.so (EKPNP_LIBRARY), e.g. other checkpoint distances / workgroup sizes of the z solve.
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
GRID="${2:-512x512x512}"
cd /tmp && export TMPDIR=/tmp
for v in $1; do
  lib="$ROOT/ek-pnp-3d_amd/libekpnp_$v.so"; [ "$v" = base ] && lib="$ROOT/ek-pnp-3d_amd/libekpnp.so"
  out="/tmp/sweep_$v"; rm -rf "$out"
  EKPNP_LIBRARY=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o t -- python3 "$ROOT/tools/time_poisson.py" "$GRID" > "$out.log" 2>&1
  echo "variant=$v: $(grep -h 'k_tridiag\|k_phi_efield' $out/t_kernel_stats.csv | awk -F, '{printf "%s avg %.1f us; ", substr($1,1,40), $4/1000}')"
done
