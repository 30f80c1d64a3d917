#!/bin/bash
# slab_overlap_ab.sh <tag> - A/B of the halo-exchange overlap knobs on the multi-rank code path with one rank
# (bench.py --force-slab: the library's RCCL transport, the ring closing on the rank itself).  Every line carries
# the `comm` block (HIP events inside libekpnp.so): halo.transfer_ms_per_step is the time from "halo buffers packed"
# to "halo landed" on the comm stream, halo.wait_ms_per_step what the compute stream still had to wait after its sweep.
set -uo pipefail
TAG="${1:-r03}"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
OUT="$ROOT/gpurun_out/slab_ab_$TAG.jsonl"
: > "$OUT"
run() {  # label, extra bench args..., env via the caller
  local label="$1"; shift
  echo "== $label" >&2
  local line
  line=$(timeout -k 10 400 python3 "$ROOT/bench.py" --force-slab --no-cpu-baseline --steps 10 --warmup 3 "$@" 2> "$ROOT/gpurun_out/slab_ab_$TAG.err" | tail -1) || { echo "FAILED: $label" >&2; tail -5 "$ROOT/gpurun_out/slab_ab_$TAG.err" >&2; return 1; }
  python3 - "$label" "$line" >> "$OUT" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
c = d.get("comm", {})
print(json.dumps({"variant": sys.argv[1], "MLUPS": d["value"], "ms_per_step": d["ms_per_step"], "in_place": d["config"]["in_place"],
                  "phases": d["config"]["phases_ms_per_step"], "halo": c.get("halo"), "edge": c.get("edge"), "phi": c.get("phi"),
                  "wait_ms_per_step": c.get("wait_ms_per_step")}))
PY
  tail -1 "$OUT" >&2
}
run "two-buffer: lead-in 2 planes, priority comm stream (round-2 default)" &&
EKPNP_SLAB_LEAD_PLANES=0 run "two-buffer: no lead-in" &&
EKPNP_COMM_CUS=8 run "two-buffer: 8 CUs kept free of the slab's kernels + lead-in" &&
EKPNP_COMM_CUS=8 EKPNP_SLAB_LEAD_PLANES=0 run "two-buffer: 8 CUs kept free, no lead-in" &&
EKPNP_COMM_CUS=16 EKPNP_SLAB_LEAD_PLANES=0 run "two-buffer: 16 CUs kept free, no lead-in" &&
EKPNP_COMM_CUS=8 EKPNP_COMM_CUS_STRICT=1 EKPNP_SLAB_LEAD_PLANES=0 run "two-buffer: 8 CUs kept free, comm stream confined to them, no lead-in" &&
run "in place: lead-in 2 planes (new)" --in-place &&
EKPNP_SLAB_LEAD_PLANES=0 run "in place: no lead-in (round-2 behaviour)" --in-place &&
EKPNP_COMM_CUS=8 EKPNP_SLAB_LEAD_PLANES=0 run "in place: 8 CUs kept free, no lead-in" --in-place
