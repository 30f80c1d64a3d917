#!/bin/bash
# ab_tri_opt.sh [grid] — rocprofv3 per-kernel time of the partition z solve (ekpnp_fast_poisson alone, tools/time_poisson.py, 33
# solves) for EKPNP_TRI_OPT = 0 (rounds 2-3), 1 (cyclic reduction stops early), 2 (chain-free pivots), 3 (both: the default).
set -uo pipefail
GRID="${1:-512x512x512}"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp
for v in 0 1 2 3; do
  OUT="$ROOT/gpurun_out/prof_tri_opt_$v"
  rm -rf "$OUT"; mkdir -p "$OUT"
  export EKPNP_TRI_OPT=$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$ROOT/tools/time_poisson.py" "$GRID" > "$OUT/trace.log" 2>&1 || { tail -5 "$OUT/trace.log"; exit 1; }
  echo "== EKPNP_TRI_OPT=$v $GRID: $(grep fast_Poisson "$OUT/trace.log" | tail -1)"
  python3 - "$OUT/trace/trace_kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_tridiag" in r["Name"] or "k_slab" in r["Name"]:
        print(f'   {r["Name"][:70]:70s} calls {r["Calls"]:>4s}  avg {float(r["AverageNs"]) / 1e3:9.1f} us  min {float(r["MinNs"]) / 1e3:9.1f}  max {float(r["MaxNs"]) / 1e3:9.1f}')
PY
  rm -rf "$OUT"
done
