"""debug: E of a 3-slab group on 48x6x388 (slabs 129/129/130) against the oracle and a single context, plane by plane"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg, O = G.load_package(), G.load_oracle()
shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "48x6x388").split("x"))
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
po = O.default_params(*shape)
orc = O.Oracle(po); orc.gpu_initialization(); start = O.perturb_fields(po, orc.fields())
orc.set_fields(start); orc.fast_poisson(); orc.init_equilibrium(); orc.step(steps); ref = orc.fields(); orc.close()
p = pkg.Params()
for name, _ in p._fields_: setattr(p, name, getattr(po, name))
def run(make):
    with make() as g:
        g.set_fields(start); g.fast_Poisson(); g.init_equilibrium(); g.step(steps)
        return {k: v.copy() for k, v in g.fields().items()}
one = run(lambda: pkg.Solver(p))
grp = run(lambda: pkg.Group(p, ns, devices=[0] * ns))
for name, f in (("single", one), ("group", grp)):
    print(name, {k: float(np.linalg.norm(f[k] - ref[k]) / (np.linalg.norm(ref[k]) + 1e-300)) for k in ("phi", "Ex", "Ey", "Ez", "c", "ux")})
for k in ("Ex", "Ey", "Ez", "phi"):
    d = np.abs(grp[k] - ref[k]).max(axis=(1, 2))
    bad = np.nonzero(d > 1e-9 * np.abs(ref[k]).max())[0]
    print(k, "planes off:", bad[:20], "max", d.max(), "scale", np.abs(ref[k]).max())
