"""Times the HIP path on the reference's default problem (50x8x51), no IO."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.load_package()
p = pkg.default_params(50, 8, 51); p.Lx, p.Ly, p.Lz = 0.5e-6, 0.08e-6, 0.5e-6
with pkg.Solver(p) as s:
    s.initialization(); s.init_equilibrium(); s.step(20); s.synchronize()
    for n in (1000, 5000):
        t0 = time.perf_counter(); s.step(n); s.synchronize(); dt = time.perf_counter() - t0
        print(f"ekpnp step on this GPU: {n} steps of 50x8x51 in {dt:.4f} s = {1e3*dt/n:.4f} ms/step = {n*20400/dt/1e6:.2f} MLUPS", flush=True)
