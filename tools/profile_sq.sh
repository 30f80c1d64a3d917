#!/bin/bash
# profile_sq.sh <tag> — SQ / LDS / L2 counters of the step's kernels on cfg3 (rocprofv3 --pmc, two
# passes with --kernel-trace only): wave counts, VALU / LDS instructions, LDS bank conflicts, issue
# stalls, L2 hit rate.  Summary -> profiles/<tag>_cfg3_pmc_sq_lds_tcc.json (tools/summarize_sq.py).
set -uo pipefail
TAG="${1:-r02}"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
OUT="$ROOT/gpurun_out/prof_sq_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --workload cfg3 --no-cpu-baseline --steps 3 --warmup 1"
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$OUT/sq" -o pmc -- $BENCH > "$OUT/sq.log" 2>&1 || { tail -5 "$OUT/sq.log"; exit 1; }
timeout -k 10 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/tcc" -o pmc -- $BENCH > "$OUT/tcc.log" 2>&1 || { tail -5 "$OUT/tcc.log"; exit 1; }
rm -f "$OUT"/*/pmc_kernel_trace.csv
du -sh "$OUT"
