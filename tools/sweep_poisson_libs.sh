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
  python3 - "$v" "$out/t_kernel_stats.csv" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[2])) if "k_tridiag" in r["Name"] or "k_phi_efield" in r["Name"]]
print("variant=%s: " % sys.argv[1] + "; ".join("%s avg %.1f us" % (r["Name"].split("(")[0].replace("void ", ""), float(r["AverageNs"]) / 1e3) for r in rows))
PY
done
