#!/bin/bash
# r05_rehearsals.sh - round 5's one-GPU-box measurements of the multi-rank code path (bench lines -> gpurun_out/r05_*.jsonl):
#  1. the one-rank ring (bench.py --force-slab) at the per-rank shapes of cfg4@8, cfg5@8 and cfg3, each line carrying the
#     after-the-fact knob A/B (`comm_ab`: defaults / inline_exchanges=0 / comm_cus=8 / lead_planes=0 / edge_chunks=4);
#  2. 2 and 4 real RCCL ranks sharing the box's one GPU (functional rehearsal of the N>1 line incl. `comm_ab`; not a bandwidth figure);
#  3. the shared-device slowdown: 4 ranks with the library's own plane transforms forced on and off,
#     with the per-stage times of the slab solve of every rank in the line.
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
O=gpurun_out
run() { echo "== $*" >&2; "$@" 2>>$O/r05_rehearsals.err; }
: > $O/r05_rehearsals.err
for wl in 512x512x128 1024x1024x128 cfg3; do
  run timeout -k 10 300 python bench.py --force-slab --workload $wl --no-cpu-baseline --steps 30 --warmup 5 >> $O/r05_one_rank_ring_comm_ab.jsonl || exit 1
done
run timeout -k 10 300 python bench.py --force-slab --workload 1024x1024x128 --in-place --no-cpu-baseline --steps 30 --warmup 5 >> $O/r05_one_rank_ring_comm_ab.jsonl || exit 1
for n in 2 4; do
  run timeout -k 10 400 python bench.py --gpus $n --scale-z 8 --single-device --steps 10 --warmup 3 --comm-ab-steps 5 >> $O/r05_rehearsal_rccl_ranks_one_gpu.jsonl || exit 1
done
run timeout -k 10 400 env EKPNP_OWN_FFT=0 python bench.py --gpus 4 --scale-z 8 --single-device --steps 10 --warmup 3 --no-comm-ab >> $O/r05_shared_device_own_fft_off_on.jsonl || exit 1
run timeout -k 10 600 env EKPNP_OWN_FFT=1 python bench.py --gpus 4 --scale-z 8 --single-device --steps 10 --warmup 3 --no-comm-ab >> $O/r05_shared_device_own_fft_off_on.jsonl || exit 1
echo done
