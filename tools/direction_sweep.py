"""direction_sweep.py [NXxNYxNZ] [in_place] - duration of the interior sweep step by step: even steps read buffer A and
write B, odd steps the reverse (in place: down / up).  On one placement the two directions can differ by several percent
(profiles/r03_placement_search_after.log); this prints both, for A/B runs of the arena knobs (EKPNP_POP_ARENA=<gap bytes>,
EKPNP_POP_CONTIGUOUS=1) in separate processes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as G
import bench

pkg = G.load_package()
grid = tuple(int(v) for v in sys.argv[1].split("x")) if len(sys.argv) > 1 else (512, 512, 512)
p = pkg.default_params(*grid)
p.in_place = int(sys.argv[2]) if len(sys.argv) > 2 else 0
prof, _ = bench.pb_profile_from_product(pkg, p)
with pkg.Solver(p) as s:
    bench.product_pb_state(s, p, prof)
    s.fast_Poisson(); s.init_equilibrium(); s.step(4); s.synchronize()
    s.kernel_timing(True)
    t = []
    for k in range(12):
        s.step(1)
        _, ms, _ = s.kernel_timing_get()
        t.append(ms)
    a, b = np.array(t[0::2]), np.array(t[1::2])
    print(f"gap {os.environ.get('EKPNP_POP_ARENA', '0'):>10} contiguous {os.environ.get('EKPNP_POP_CONTIGUOUS', '0')}: sweep {a.mean():7.3f} / {b.mean():7.3f} ms in the two directions "
          f"(min {a.min():.3f} / {b.min():.3f}), mean {0.5 * (a.mean() + b.mean()):7.3f}, placement {s.placement_report()}", flush=True)
