#!/bin/bash
# r05_rehearsals2.sh - after the hardware-queue finding: the 2- and 4-rank rehearsals of the N>1 line (with comm_ab) under
# GPU_MAX_HW_QUEUES=1 (bench.py --single-device sets it), and the one configuration of the experiment log that printed no line.
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
O=gpurun_out
: > $O/r05_rehearsals2.err
rm -f $O/r05_rehearsal_rccl_ranks_one_gpu_hwq1.jsonl
for n in 2 4; do
  echo "== $n ranks" >> $O/r05_rehearsals2.err
  timeout -k 10 400 python bench.py --gpus $n --scale-z 8 --single-device --steps 10 --warmup 3 --comm-ab-steps 5 >> $O/r05_rehearsal_rccl_ranks_one_gpu_hwq1.jsonl 2>> $O/r05_rehearsals2.err || exit 1
done
echo "== cfg5 / 16, 4 ranks" >> $O/r05_rehearsals2.err
timeout -k 10 400 python bench.py --gpus 4 --workload cfg5 --scale-z 16 --single-device --steps 6 --warmup 2 --comm-ab-steps 3 >> $O/r05_rehearsal_rccl_ranks_one_gpu_hwq1.jsonl 2>> $O/r05_rehearsals2.err || exit 1
echo "== inline_exchanges=0 from the environment, 4 ranks, 4 hardware queues per process (the line the experiment log lacks)" >> $O/r05_rehearsals2.err
timeout -k 10 300 env EKPNP_INLINE_EXCHANGES=0 GPU_MAX_HW_QUEUES=4 python bench.py --gpus 4 --scale-z 8 --single-device --steps 8 --warmup 3 --no-comm-ab > $O/r05_inline0_hwq4.json 2> $O/r05_inline0_hwq4.err; echo "rc=$?" >> $O/r05_rehearsals2.err
echo done
