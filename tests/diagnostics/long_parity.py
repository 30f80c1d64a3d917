"""Diagnostic: growth of the HIP-vs-oracle difference over a long run (16x12x17, 4 lattices)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as G
pkg = G.load_package(); O = G.load_oracle()
po = O.default_params(16, 12, 17); po.pb_iterations = 60
p = pkg.Params()
for n, _ in p._fields_: setattr(p, n, getattr(po, n))
orc = O.Oracle(po); orc.initialization()
start = O.perturb_fields(po, orc.fields())
orc.set_fields(start); orc.fast_poisson(); orc.init_equilibrium()
with pkg.Solver(p) as s:
    s.initialization(); s.set_fields(start); s.fast_Poisson(); s.init_equilibrium()
    done = 0
    for mark in (10, 100, 500, 1000, 2000, 4000):
        orc.step(mark - done); s.step(mark - done); done = mark
        e = O.rel_l2(s.fields(), orc.fields())
        print(mark, {k: float(f"{v:.1e}") for k, v in e.items()}, "umax", float(np.abs(orc.field("uz")).max()), flush=True)
