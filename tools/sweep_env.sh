#!/bin/bash
# sweep_env.sh VAR "v1 v2 ..." [workload] — A/B an environment tuning knob on one box
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
VAR="$1"; VALS="$2"; WL="${3:-cfg3}"
for rep in 1 2; do
for v in $VALS; do
  echo -n "$VAR=$v rep=$rep: "
  env $VAR=$v timeout -k 10 300 python3 $ROOT/bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline --ic uniform 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], 'MLUPS', d['ms_per_step'], 'ms/step; bulk', d['roofline']['avg_launch_ms'], 'ms')
"
done
done
