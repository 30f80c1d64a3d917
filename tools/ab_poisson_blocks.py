"""A/B of ekpnp_tune(ctx, "poisson_blocks", n) - the middle passes of the single context's solve taken kx block by kx block.

Per grid: random c, cn; phi of the one-block solve is the yardstick; every block count must return the same bits;
ms per solve over 30 solves (k_poisson_rhs included in every leg alike).  Usage: ab_poisson_blocks.py [grid ...] [--blocks 1,2,4,...]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G  # noqa: E402

pkg = G.load_package()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
blocks = [1, 2, 3, 4, 6, 8, 11, 16, 33, 1]
zchunks = [0]
for a in sys.argv[1:]:
    if a.startswith("--blocks="):
        blocks = [int(v) for v in a.split("=", 1)[1].split(",")]
    if a.startswith("--zchunks="):  # planes per chunk of the row + column passes ("poisson_zchunk"), 0 = whole passes
        zchunks = [int(v) for v in a.split("=", 1)[1].split(",")]
grids = args or ["512x512x512"]
for g in grids:
    shape = tuple(int(v) for v in g.split("x"))
    p = pkg.default_params(*shape)
    p.n_lattices = 1  # the solve does not depend on the populations: keep the footprint small
    p.chargeinf = 0.0
    p.Ra = 0.0
    s = pkg.Solver(p)
    rng = np.random.default_rng(5)
    zyx = (shape[2], shape[1], shape[0])
    s.set_field("c", 0.01 * (1.0 + 0.1 * rng.standard_normal(zyx)))
    s.set_field("cn", 0.01 * (1.0 + 0.1 * rng.standard_normal(zyx)))
    ref = None
    for nb, zc in [(b, z) for z in zchunks for b in blocks]:
        s.tune("poisson_blocks", nb)
        s.tune("poisson_zchunk", zc)
        for _ in range(3):
            s.fast_Poisson()
        s.synchronize()
        n = 30
        t = time.perf_counter()
        for _ in range(n):
            s.fast_Poisson()
        s.synchronize()
        dt = (time.perf_counter() - t) / n
        phi = s.get_field("phi")
        if ref is None:
            ref = phi
        same = bool(np.array_equal(ref, phi))
        print(json.dumps({"grid": g, "poisson_blocks": nb, "poisson_zchunk": zc, "ms_per_solve": round(dt * 1e3, 4), "same_bits_as_one_block": same,
                          "phi_absmax": float(np.abs(phi).max())}), flush=True)
        del phi
    s.close()
