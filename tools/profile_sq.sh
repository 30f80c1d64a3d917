#!/bin/bash
# profile_sq.sh <tag> [label] [bench args...] — SQ / LDS / L2 counters of the step's kernels (rocprofv3 --pmc, separate
# passes with --kernel-trace only): wave counts and cycles, VALU / LDS / VMEM instructions, LDS bank conflicts, issue
# stalls, L2 hit rate.  Default workload: cfg3.  Summary -> gpurun_out/<tag>_<label>_pmc_sq_lds_tcc.json (tools/summarize_sq.py, run on the box; copy to profiles/).
# With EKPNP_PROFILE_HBM=1 two more passes collect FETCH_SIZE / WRITE_SIZE (HBM bytes per launch) of the same command.
set -uo pipefail
TAG="${1:-r04}"; shift || true
LABEL="${1:-cfg3}"; shift || true
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
OUT="$ROOT/gpurun_out/prof_sq_${TAG}_${LABEL}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
if [ "$#" -gt 0 ]; then ARGS="$*"; else ARGS="--workload cfg3"; fi
BENCH="python3 $ROOT/bench.py $ARGS --no-cpu-baseline --no-batch-ab --no-comm-ab --steps 3 --warmup 1"  # (no after-the-fact A/B legs: they would dilute the per-kernel means)
pass() {  # <dir> <counters...>
  local d="$1"; shift
  timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$d" -o pmc -- $BENCH > "$OUT/$d.log" 2>&1 || { tail -5 "$OUT/$d.log"; exit 1; }
  rm -f "$OUT/$d"/pmc_kernel_trace.csv
}
pass sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass tcc TCC_HIT_sum TCC_MISS_sum
if [ "${EKPNP_PROFILE_HBM:-0}" = "1" ]; then
  pass fetch FETCH_SIZE
  pass write WRITE_SIZE
fi
python3 "$ROOT/tools/summarize_sq.py" "$TAG" "$LABEL" > "$OUT.summary.log" 2>&1 || { tail -5 "$OUT.summary.log"; exit 1; }
rm -rf "$OUT"   # per-dispatch counter CSVs: tens of MB each, more than gpurun brings back
tail -n 60 "$OUT.summary.log"
