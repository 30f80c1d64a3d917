#!/bin/bash
# ab_own_fft.sh [grid] — rocprofv3 per-kernel times of ekpnp_fast_poisson alone with rocFFT plans (EKPNP_OWN_FFT=0) and with the
# library's own row / column passes (EKPNP_OWN_FFT=1), tools/time_poisson.py, 33 solves.
set -uo pipefail
GRID="${1:-512x512x512}"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  OUT="$ROOT/gpurun_out/prof_own_fft_$v"
  rm -rf "$OUT"; mkdir -p "$OUT"
  export EKPNP_OWN_FFT=$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$ROOT/tools/time_poisson.py" "$GRID" > "$OUT/trace.log" 2>&1 || { tail -5 "$OUT/trace.log"; exit 1; }
  echo "== EKPNP_OWN_FFT=$v $GRID: $(grep fast_Poisson "$OUT/trace.log" | tail -1)"
  python3 - "$OUT/trace/trace_kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["Calls"]) >= 30 and float(r["AverageNs"]) > 20000:
        print(f'   {r["Name"][:84]:84s} calls {r["Calls"]:>4s}  avg {float(r["AverageNs"]) / 1e3:9.1f} us')
PY
  rm -rf "$OUT"
done
