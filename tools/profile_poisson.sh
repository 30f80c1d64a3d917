#!/bin/bash
# profile_poisson.sh <tag> [grid] — rocprofv3 per-kernel times of ekpnp_fast_poisson alone
# (tools/time_poisson.py: 33 solves on one grid, populations of one lattice only).
set -uo pipefail
TAG="${1:-r02}"
GRID="${2:-512x512x512}"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
OUT="$ROOT/gpurun_out/prof_poisson_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$ROOT/tools/time_poisson.py" "$GRID" > "$OUT/trace.log" 2>&1 || { tail -5 "$OUT/trace.log"; exit 1; }
tail -2 "$OUT/trace.log"
rm -f "$OUT/trace/trace_kernel_trace.csv"
cut -c1-150 "$OUT/trace/trace_kernel_stats.csv" | head -12
