"""Times the phases of the HIP path on one grid (diagnostic)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.load_package()
shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "50x8x51").split("x"))
t = time.perf_counter()
def lap(msg):
    global t
    n = time.perf_counter(); print(f"{msg}: {n - t:.3f} s", flush=True); t = n
p = pkg.default_params(*shape); p.pb_iterations = 20
s = pkg.Solver(p); lap("create (plans)")
s.initialization(); s.synchronize(); lap("initialization 20 sweeps")
s.initialization(); s.synchronize(); lap("initialization 20 sweeps again")
s.init_equilibrium(); s.synchronize(); lap("init_equilibrium")
s.step(1); s.synchronize(); lap("first step")
s.step(20); s.synchronize(); lap("20 steps")
for i in range(20): s.fast_Poisson()
s.synchronize(); lap("20 poisson")
for i in range(20): s.stream_collide_save()
s.synchronize(); lap("20 lbm")
f = s.fields(); lap("fields D2H")
s.close(); lap("close")
s2 = pkg.Solver(p); lap("second create")
s2.close()
