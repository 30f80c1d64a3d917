#!/bin/bash
# profile.sh <tag> [workload] — rocprofv3 passes of bench.py on the GPU box (run through gpurun):
#   1. --kernel-trace --stats          (per-kernel time)        -> gpurun_out/prof_<tag>/trace
#   2. --pmc FETCH_SIZE, --pmc WRITE_SIZE (own passes)           -> gpurun_out/prof_<tag>/pmc_*
#   3. the same two counters on tools/pmc_calib (known bytes)    -> gpurun_out/prof_<tag>/calib_*
# Counters are collected in their own runs with --kernel-trace only (no sys/hip/hsa tracing).
set -uo pipefail
TAG="${1:-r01}"
WL="${2:-cfg3}"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
OUT="$ROOT/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --workload $WL --no-cpu-baseline --no-batch-ab --no-comm-ab"  # (the after-the-fact A/B legs would dilute the per-kernel means)
echo "== kernel trace" && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH --steps 10 --warmup 2 > "$OUT/trace.log" 2>&1 || exit 1
echo "== pmc FETCH_SIZE" && timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o pmc -- $BENCH --steps 3 --warmup 1 > "$OUT/pmc_fetch.log" 2>&1 || exit 1
echo "== pmc WRITE_SIZE" && timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o pmc -- $BENCH --steps 3 --warmup 1 > "$OUT/pmc_write.log" 2>&1 || exit 1
echo "== calib FETCH_SIZE" && timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/calib_fetch" -o pmc -- "$ROOT/tools/pmc_calib" > "$OUT/calib_fetch.log" 2>&1 || exit 1
echo "== calib WRITE_SIZE" && timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/calib_write" -o pmc -- "$ROOT/tools/pmc_calib" > "$OUT/calib_write.log" 2>&1 || exit 1
find "$OUT" -name "*.csv" | head -40
# keep only what is needed (<64 MiB merge limit): drop per-dispatch traces of the big run
du -sh "$OUT"
