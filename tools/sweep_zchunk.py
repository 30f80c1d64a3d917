"""sweep_zchunk.py [reps] - planes per launch of the two-buffer collide sweep on cfg3, ONE context,
every value timed `reps` times in turn (ms per step over 30 steps, bulk-kernel ms from the library's
HIP events).  A single launch of the whole sweep shows a run-to-run spread of +-1.3 % on one box;
launches of a few dozen planes re-align the eight XCDs and are steadier."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as G
import bench

pkg = G.load_package()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
values = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,8,16,32,64,128,255".split(","))]
p = pkg.default_params(512, 512, 512)
prof, _ = bench.pb_profile_from_product(pkg, p)
s = pkg.Solver(p)
bench.product_pb_state(s, p, prof)
bench.apply_perturbation(s, None, p)
s.fast_Poisson(); s.init_equilibrium(); s.step(10); s.synchronize()
res = {v: [] for v in values}
for rep in range(reps):
    for v in values:
        s.tune("ab_zchunk", v)
        s.step(3); s.synchronize()
        s.kernel_timing(True)
        t0 = time.perf_counter(); s.step(30); s.synchronize(); dt = time.perf_counter() - t0
        n, ms, nodes = s.kernel_timing_get(); s.kernel_timing(False)
        res[v].append((dt / 30 * 1e3, ms / 30))
        print(f"rep {rep} zchunk {v:4d}: {dt / 30 * 1e3:8.3f} ms/step, bulk {ms / 30:8.3f} ms", flush=True)
print("\nmean over reps (ms/step, bulk ms, MLUPS), min..max of ms/step:")
for v in values:
    a = np.array(res[v])
    print(f"zchunk {v:4d}: {a[:, 0].mean():8.3f} {a[:, 1].mean():8.3f} {512**3 / a[:, 0].mean() / 1e3:8.1f}   {a[:, 0].min():.3f}..{a[:, 0].max():.3f}")
s.close()
