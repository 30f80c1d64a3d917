"""Shared by tests/test_parity_gpu.py (cfg3, two buffers) and tests/test_group_gpu.py (cfg4's lattice in place): full-size
parity with x-y STRUCTURED data through a size-independent property."""
import numpy as np
import pytest


def periodic_tile_check(pkg, O, nz, in_place, steps, min_free_bytes, index_bits):
    """A 64 x 64 tile with full x-y structure (the bench's Gouy-Chapman walls + the closed-form 3-D perturbation of SURVEY
    8(c) with period 64), replicated 8 x 8 over 512 x 512 planes, is a periodic problem that does not know how often it is
    repeated: after `steps` steps EVERY node of the 512 x 512 x nz lattice must equal the node of the 64 x 64 x nz run it
    is a copy of (only the FFT sizes, i.e. rounding, differ).  Unlike an x-y uniform start this sees a swapped / mirrored
    direction, a wrong tile offset and an index that wraps: all 64 tiles are compared, so the ones at opposite corners and
    the one holding population element 2^index_bits are among them."""
    import torch

    import bench

    free_b, _ = torch.cuda.mem_get_info()
    if free_b < min_free_bytes:
        pytest.skip(f"needs {min_free_bytes / 1e9:.0f} GB of free HBM")
    names = ("rho", "c", "cn", "T", "phi", "ux", "uy", "uz", "Ez")
    p = pkg.default_params(64, 64, nz)
    p.in_place = in_place
    with pkg.Solver(p) as s:
        bench.gouy_chapman_state(s, p)
        start = O.perturb_fields(p, {k: s.get_field(k) for k in ("rho", "c", "cn", "T", "ux", "uy", "uz")})
        s.set_fields(start)
        s.fast_Poisson(); s.init_equilibrium(); s.step(steps)
        small = {k: s.get_field(k) for k in names}
    p = pkg.default_params(512, 512, nz)
    p.in_place = in_place
    worst = {}
    with pkg.Solver(p) as s:
        # the far end of the index space is really there: elements of one population buffer beyond 2^index_bits
        planes = nz + 2 + ((min(nz // 4, 64) + 1) if in_place else 0)
        assert planes * 512 * 8 * 27 * 64 > 2**index_bits
        bench.gouy_chapman_state(s, p)
        for k, v in start.items():
            s.set_field(k, np.tile(v, (1, 8, 8)))
        s.fast_Poisson(); s.init_equilibrium(); s.step(steps)
        for k in names:
            big = s.get_field(k)
            big -= np.tile(small[k], (1, 8, 8))
            worst[k] = float(np.abs(big).max() / np.abs(small[k]).max())
            del big
    bad = {k: v for k, v in worst.items() if not v <= (1e-7 if k in ("ux", "uy", "uz") else 1e-11)}
    assert not bad, worst
    return worst
