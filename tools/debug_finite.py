"""Diagnostic: where does a run stop being finite?"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import __graft_entry__ as G
pkg = G.load_package()
def chk(s, tag):
    bad = []
    for k in pkg.FIELDS:
        a = s.get_field(k)
        if not np.isfinite(a).all():
            idx = np.argwhere(~np.isfinite(a))
            bad.append((k, len(idx), idx[0].tolist(), idx[-1].tolist()))
    print(tag, "OK" if not bad else bad, flush=True)
    return not bad
for shape, nl in (((256, 64, 66), 4), ((256, 256, 64), 3), ((256, 256, 256), 3)):
    print("====", shape, nl, flush=True)
    p = pkg.default_params(*shape); p.n_lattices = nl; p.pb_iterations = 30
    if nl < 4: p.Ra = 0.0
    s = pkg.Solver(p)
    s.initialization(); chk(s, "after initialization")
    bench.apply_perturbation(s, None, p); chk(s, "after perturbation")
    s.fast_Poisson(); chk(s, "after poisson")
    s.init_equilibrium()
    for k in range(1, 8):
        s.stream_collide_save(); ok1 = chk(s, f"step {k} lbm")
        s.fast_Poisson(); ok2 = chk(s, f"step {k} poisson")
        if not (ok1 and ok2): break
    s.close()
