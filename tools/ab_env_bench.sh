#!/bin/bash
# ab_env_bench.sh <out.jsonl> <bench args in one string> "A=1 B=2" "A=3" ... — A/B of environment knobs inside the full bench
# step (same box, legs in the order given; repeat a setting to alternate): one short bench.py line per leg
out=$1; bargs=$2; shift 2
: > "$out"
for v in "$@"; do
  env $v timeout -k 10 200 python bench.py $bargs --no-cpu-baseline --no-batch-ab --no-comm-ab 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'env': '$v', 'bench_args': '$bargs', 'MLUPS': d['value'], 'ms_per_step': d['ms_per_step'], 'phases': d['config']['phases_ms_per_step'], 'finite': d['config']['finite']}))" >> "$out" || exit 1
done
