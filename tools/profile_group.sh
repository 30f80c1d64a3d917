#!/bin/bash
# profile_group.sh <tag> [grid] [nslabs] - rocprofv3 kernel trace of ekpnp_group_step on `nslabs` in-place slabs on device 0:
# per-kernel time of thin slabs, and the device's idle fraction during the timed steps (tools/gpu_busy_from_trace.py) -
# the measure of whether ONE host thread keeps the device(s) fed.
set -uo pipefail
TAG="${1:-r03}"; GRID="${2:-512x512x768}"; NS="${3:-8}"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
OUT="$ROOT/gpurun_out/prof_group_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$ROOT/tools/group_overhead.py" "$GRID" "$NS" 10 --group-only > "$OUT/trace.log" 2>&1 || { tail -5 "$OUT/trace.log"; exit 1; }
T=$(find "$OUT" -name "*kernel_trace.csv" | head -1)
S=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
python3 "$ROOT/tools/gpu_busy_from_trace.py" "$T" 8 > "$ROOT/gpurun_out/${TAG}_group_gpu_busy.json" && cp "$S" "$ROOT/gpurun_out/${TAG}_group_kernel_stats.csv"
tail -1 "$OUT/trace.log" >> "$ROOT/gpurun_out/${TAG}_group_gpu_busy.json"
rm -rf "$OUT/trace"
cat "$ROOT/gpurun_out/${TAG}_group_gpu_busy.json"
