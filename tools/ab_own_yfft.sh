#!/bin/bash
# A/B on cfg3: rocFFT's 2-D plan (EKPNP_OWN_YFFT=0) vs rocFFT rows + the library's own y pass (default).
# usage (GPU box): bash tools/ab_own_yfft.sh > gpurun_out/ab_own_yfft.log
for v in 0 1 0 1; do
  EKPNP_OWN_YFFT=$v timeout -k 10 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null \
    | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('own_yfft', '$v', 'MLUPS', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['config']['step_roofline_frac'], 'phases', d['config']['phases_ms_per_step'])"
done
