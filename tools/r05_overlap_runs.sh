#!/bin/bash
# r05_overlap_runs.sh - regenerate the halo-overlap traces at HEAD: the multi-rank code path on one
# rank (ring to itself) under rocprofv3 --kernel-trace, per-rank shapes of cfg4@8 (512x512x128), cfg5@8 (1024x1024x128)
# and cfg3 through the slab path, two population buffers and in place.  One rocprofv3 run after the other, never two.
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
bash tools/profile_slab.sh r05_512x512x128 --workload 512x512x128 &&
bash tools/profile_slab.sh r05_512x512x128_in_place --workload 512x512x128 --in-place &&
bash tools/profile_slab.sh r05_cfg5_rank_shape --workload 1024x1024x128 &&
bash tools/profile_slab.sh r05_cfg5_rank_shape_in_place --workload 1024x1024x128 --in-place &&
bash tools/profile_slab.sh r05_cfg3 --workload cfg3 &&
bash tools/profile_slab.sh r05_cfg3_in_place --workload cfg3 --in-place
