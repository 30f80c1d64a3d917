#!/bin/bash
# bench_round.sh <tag> — the round's bench lines on one box: the driver-style cfg3 line, three short repeats, the secondary
# workloads (cfg2, cfg3 in place, 512x512x1024 in place) and the per-rank slab shapes of cfg4@4, cfg4@8 and cfg5@8 through the
# multi-rank code path on one rank (library RCCL transport, ring to itself).  Output: gpurun_out/<tag>_bench_*.json(l).
set -uo pipefail
TAG="${1:-r04}"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"; mkdir -p gpurun_out
python bench.py > "gpurun_out/${TAG}_bench_cfg3.json" 2> "gpurun_out/${TAG}_bench_cfg3.err" || { tail -3 "gpurun_out/${TAG}_bench_cfg3.err"; exit 1; }
: > "gpurun_out/${TAG}_bench_cfg3_repeats.jsonl"
for i in 1 2 3; do python bench.py --steps 25 --warmup 5 --no-cpu-baseline 2>/dev/null >> "gpurun_out/${TAG}_bench_cfg3_repeats.jsonl"; done
: > "gpurun_out/${TAG}_bench_secondary.jsonl"
python bench.py --workload cfg2 --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null >> "gpurun_out/${TAG}_bench_secondary.jsonl"
python bench.py --in-place --steps 25 --warmup 5 --no-cpu-baseline 2>/dev/null >> "gpurun_out/${TAG}_bench_secondary.jsonl"
python bench.py --in-place --workload 512x512x1024 --steps 15 --warmup 3 --no-cpu-baseline 2>/dev/null >> "gpurun_out/${TAG}_bench_secondary.jsonl"
: > "gpurun_out/${TAG}_bench_slab_shapes.jsonl"
python bench.py --force-slab --steps 25 --warmup 5 --no-cpu-baseline 2>/dev/null >> "gpurun_out/${TAG}_bench_slab_shapes.jsonl"
python bench.py --force-slab --workload 512x512x256 --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null >> "gpurun_out/${TAG}_bench_slab_shapes.jsonl"
python bench.py --force-slab --workload 512x512x128 --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null >> "gpurun_out/${TAG}_bench_slab_shapes.jsonl"
python bench.py --force-slab --workload 1024x1024x128 --steps 25 --warmup 5 --no-cpu-baseline 2>/dev/null >> "gpurun_out/${TAG}_bench_slab_shapes.jsonl"
python3 - "$TAG" <<'PY'
import json, sys
tag = sys.argv[1]
for f in (f"gpurun_out/{tag}_bench_cfg3.json", f"gpurun_out/{tag}_bench_cfg3_repeats.jsonl", f"gpurun_out/{tag}_bench_secondary.jsonl", f"gpurun_out/{tag}_bench_slab_shapes.jsonl"):
    for l in open(f):
        if l.strip().startswith("{"):
            d = json.loads(l)
            print(f'{d["value"]:9.1f} MLUPS {d["ms_per_step"]:8.3f} ms  {d["config"]["workload"][:70]:70s} {d["config"]["phases_ms_per_step"]} step_frac {d["config"]["step_roofline_frac"]} bulk_frac {d["roofline"]["frac"]} dev_GB {d["config"]["device_bytes"] / 1e9:.1f}')
PY
