"""Times ekpnp_fast_poisson alone on one grid (diagnostic): ms per solve over 30 solves."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G

pkg = G.load_package()
shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "512x512x512").split("x"))
p = pkg.default_params(*shape)
p.n_lattices = 1  # the Poisson solve does not depend on the populations: keep the footprint small
p.chargeinf = 0.0
p.Ra = 0.0
s = pkg.Solver(p)
for _ in range(3):
    s.fast_Poisson()
s.synchronize()
t = time.perf_counter()
n = 30
for _ in range(n):
    s.fast_Poisson()
s.synchronize()
dt = (time.perf_counter() - t) / n
nodes = shape[0] * shape[1] * shape[2]
print(f"{shape[0]}x{shape[1]}x{shape[2]}: fast_Poisson {dt * 1e3:.3f} ms  ({nodes * 48 / dt / 1e9:.0f} GB/s of the 48 B/node it must move)", flush=True)
s.close()
