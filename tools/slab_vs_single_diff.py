"""How far is a one-rank slab (library transport, ring to itself) from a single context after a few steps?  Max abs / rel
difference per field, for the knob combinations given as KEY=VAL,... on the command line (each combination a child process,
because the knobs are read at creation / first use).
    python tools/slab_vs_single_diff.py [NXxNYxNZ] [steps] "EKPNP_LAZY_E=0" "EKPNP_HALO_DIRECT=0,EKPNP_LAZY_E=0" ... """
import os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as G
import bench
pkg = G.load_package()
nx, ny, nz = (int(v) for v in sys.argv[1].split("x")); steps = int(sys.argv[2])
p = pkg.default_params(nx, ny, nz); p.pb_iterations = 8
def drive(s):
    s.initialization()
    bench.apply_perturbation(s, None, p)
    s.fast_Poisson(); s.init_equilibrium(); s.step(steps)
    return s.fields()
with pkg.Solver(p) as s:
    a = drive(s)
s = pkg.Solver(p, 0, 1, slab=True); s.attach_comm(pkg.comm_unique_id()); b = drive(s); s.close()
print({k: (float(np.abs(a[k] - b[k]).max()), float(np.abs(a[k] - b[k]).max() / (np.abs(a[k]).max() + 1e-300))) for k in a})
''' % ROOT
grid = sys.argv[1] if len(sys.argv) > 1 else "20x6x24"
steps = sys.argv[2] if len(sys.argv) > 2 else "5"
for combo in (sys.argv[3:] or [""]):
    env = dict(os.environ)
    for kv in filter(None, combo.split(",")):
        k, v = kv.split("=")
        env[k] = v
    r = subprocess.run([sys.executable, "-c", CHILD, grid, steps], env=env, capture_output=True, text=True)
    print(f"== {combo or 'defaults'}\n{r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-1500:]}", flush=True)
