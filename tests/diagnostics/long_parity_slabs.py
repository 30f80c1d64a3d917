"""Diagnostic: a long run on z slabs - 4 slabs (device-copy transport) and 1 slab over RCCL against the
single-context run and the oracle, 24x8x33 (uneven slabs), 4 lattices, up to 3000 steps."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as G
pkg = G.load_package(); O = G.load_oracle()
shape = (24, 8, 33)
po = O.default_params(*shape); po.pb_iterations = 40
p = pkg.Params()
for n, _ in p._fields_: setattr(p, n, getattr(po, n))
orc = O.Oracle(po); orc.initialization()
start = O.perturb_fields(po, orc.fields())
orc.set_fields(start); orc.fast_poisson(); orc.init_equilibrium()
one = pkg.Solver(p); four = pkg.Group(p, 4, devices=[0] * 4); ring = pkg.Group(p, 1, devices=[0], transport=pkg.TRANSPORT_RCCL)
for r in (one, four, ring):
    r.initialization(); r.set_fields(start); r.fast_Poisson(); r.init_equilibrium()
done = 0
for mark in (10, 100, 500, 1500, 3000):
    orc.step(mark - done)
    for r in (one, four, ring): r.step(mark - done)
    done = mark
    f1 = one.fields()
    fmt = lambda e: {k: float(f"{v:.1e}") for k, v in e.items()}
    print(mark, "one vs oracle", fmt(O.rel_l2(f1, orc.fields())), flush=True)
    print(mark, "4 slabs vs one", fmt(O.rel_l2(four.fields(), f1)), flush=True)
    print(mark, "rccl ring vs one", fmt(O.rel_l2(ring.fields(), f1)), "current", one.current(), four.current(), flush=True)
for r in (one, four, ring): r.close()
