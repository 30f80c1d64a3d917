"""A/B on cfg3-sized planes: serial Thomas sweeps (tri_partition 0) vs the partition solve (1).  33 Poisson solves each,
HIP-event phase timing of the library.  usage (GPU box): python tools/ab_tri_partition.py [NXxNYxNZ]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

pkg = G.load_package()
nx, ny, nz = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "512x512x512").split("x"))
p = pkg.default_params(nx, ny, nz)
p.n_lattices = 1  # populations are not needed for the solve
p.chargeinf, p.Ra, p.TH = 0.0, 0.0, 0.0
with pkg.Solver(p) as s:
    rng = np.random.default_rng(0)
    s.set_field("c", 0.01 * (1 + 0.1 * rng.random((nz, ny, nx))))
    s.set_field("cn", 0.01 * (1 + 0.1 * rng.random((nz, ny, nx))))
    ref = None
    for knob in (0, 1, 0, 1):
        s.tune("tri_partition", knob)
        for _ in range(3):
            s.fast_Poisson()
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            s.fast_Poisson()
        s.synchronize()
        dt = (time.perf_counter() - t0) / 30
        phi = s.get_field("phi")
        if ref is None:
            ref = phi
        print(f"tri_partition={knob}: fast_Poisson {dt * 1e3:.3f} ms (with k_poisson_rhs), max|phi - first| / max|phi| = {np.abs(phi - ref).max() / np.abs(ref).max():.2e}", flush=True)
