#!/bin/bash
# sweep.sh — A/B the tuning knobs of the bulk kernel on one box (same process conditions).
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
WL="${1:-cfg3}"
for rep in 1 2; do
for m in ${MAPS:-0 1 2}; do
  echo -n "map=$m rep=$rep: "
  EKPNP_BULK_MAP=$m timeout -k 10 300 python3 $ROOT/bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline --ic uniform 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], 'MLUPS', d['ms_per_step'], 'ms/step; bulk', d['roofline']['avg_launch_ms'], 'ms', d['roofline']['achieved'], 'GB/s')
"
done
done
