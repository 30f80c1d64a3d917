#!/bin/bash
# A/B of EKPNP_POISSON_BLOCKS inside the full cfg3 step (same box, alternating legs): bench.py lines, one per leg
out=${1:-gpurun_out/r05c_ab_poisson_blocks_bench.jsonl}
: > "$out"
for nb in 1 3 4 6 1 3 4 6; do
  EKPNP_POISSON_BLOCKS=$nb timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-batch-ab 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'poisson_blocks': $nb, 'MLUPS': d['value'], 'ms_per_step': d['ms_per_step'], 'phases': d['config']['phases_ms_per_step']}))" >> "$out" || exit 1
done
