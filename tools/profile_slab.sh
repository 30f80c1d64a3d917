#!/bin/bash
# profile_slab.sh <tag> [bench args...] — rocprofv3 kernel trace of the multi-rank code path on one rank
# (bench.py --force-slab: slab context + the library's RCCL transport, ring closing on itself).
# The per-dispatch trace (start/end timestamps) shows where the RCCL kernel of the halo exchange
# sits relative to k_collide_bulk of the interior planes -> tools/overlap_trace.py.
# Environment knobs (EKPNP_COMM_CUS, EKPNP_SLAB_LEAD_PLANES ...) pass through from the caller.
set -uo pipefail
TAG="${1:-r03}"; shift || true
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
OUT="$ROOT/gpurun_out/prof_slab_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$ROOT/bench.py" --force-slab --no-cpu-baseline --no-batch-ab --no-comm-ab --steps 8 --warmup 2 "$@" > "$OUT/trace.log" 2>&1 || { tail -5 "$OUT/trace.log"; exit 1; }
T=$(find "$OUT" -name "*kernel_trace.csv" | head -1)
S=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
python3 "$ROOT/tools/overlap_trace.py" "$T" > "$ROOT/gpurun_out/${TAG}_slab_overlap.json" || { echo "overlap_trace found no step: stale tool?"; exit 1; }
cp "$S" "$ROOT/gpurun_out/${TAG}_slab_kernel_stats.csv"
python3 "$ROOT/tools/step_timeline.py" "$T" > "$ROOT/gpurun_out/${TAG}_slab_last_step_timeline.json" || true
python3 "$ROOT/tools/trace_steps.py" --excerpt "$T" "$ROOT/gpurun_out/${TAG}_slab_trace_excerpt_last_3_steps.csv" 3 || true
grep '"metric"' "$OUT/trace.log" | tail -1 > "$ROOT/gpurun_out/${TAG}_slab_bench_line_under_profiler.json"
rm -rf "$OUT/trace"   # the per-dispatch trace is large; the three summaries above are what is kept
head -c 1500 "$ROOT/gpurun_out/${TAG}_slab_overlap.json"
