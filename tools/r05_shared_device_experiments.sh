#!/bin/bash
# r05_shared_device_experiments.sh - what makes 4 ranks on ONE device 10-30x slower with the own plane transforms?
# One 4-rank rehearsal (bench.py --gpus 4 --scale-z 8 --single-device) per environment; every line keeps the per-rank
# stage times of the slab solve.  Functional box, never a scaling figure.
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
OUT=gpurun_out/r05_shared_device_experiments.log
: > $OUT
one() {
  echo "== $*" >> $OUT
  timeout -k 10 300 env "$@" python bench.py --gpus 4 --scale-z 8 --single-device --steps 8 --warmup 3 --no-comm-ab 2>/dev/null | python3 -c '
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
c = d["config"]
print(d["value"], "MLUPS", d["ms_per_step"], "ms/step", c["plane_transforms"], "phases max", c["phases_ms_per_step_by_rank"]["max"], "stages min", c["poisson_stages_ms_per_solve_by_rank"]["min"], "max", c["poisson_stages_ms_per_solve_by_rank"]["max"])' >> $OUT 2>&1 || echo "FAILED" >> $OUT
}
one EKPNP_OWN_FFT=0
one X=auto
one EKPNP_OWN_FFT=1
one EKPNP_OWN_FFT=1 EKPNP_ALSO_MAKE_PLANS=1
one EKPNP_OWN_FFT=1 GPU_MAX_HW_QUEUES=1
one EKPNP_OWN_FFT=1 GPU_MAX_HW_QUEUES=8
one EKPNP_OWN_FFT=0 GPU_MAX_HW_QUEUES=1
one EKPNP_OWN_FFT=1 EKPNP_INLINE_EXCHANGES=0
one EKPNP_OWN_FFT=0
cat $OUT
