#!/bin/bash
# profile_slab.sh <tag> — rocprofv3 kernel trace of the multi-rank code path on one rank
# (bench.py --force-slab: slab context + the library's RCCL transport, ring closing on itself).
# The per-dispatch trace (start/end timestamps) shows where the RCCL kernel of the halo exchange
# sits relative to k_collide_bulk of the interior planes -> tools/overlap_trace.py.
set -uo pipefail
TAG="${1:-r02}"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
OUT="$ROOT/gpurun_out/prof_slab_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$ROOT/bench.py" --force-slab --no-cpu-baseline --steps 8 --warmup 2 > "$OUT/trace.log" 2>&1 || { tail -5 "$OUT/trace.log"; exit 1; }
find "$OUT" -name "*.csv" | head; du -sh "$OUT"
