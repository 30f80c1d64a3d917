"""placement_spread.py [n] - step time of cfg3 in n contexts created one after the other in ONE process
(each is destroyed before the next is made).  Within a context the step time repeats to 0.01 %; between
contexts (and between processes) it moves by up to 2.5 %: where hipMalloc places the arrays matters."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as G
import bench

pkg = G.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
grid = tuple(int(v) for v in sys.argv[3].split("x")) if len(sys.argv) > 3 else (512, 512, 512)  # round 3: thin lattices too
p = pkg.default_params(*grid)
if len(sys.argv) > 2:
    p.in_place = int(sys.argv[2])
prof, _ = bench.pb_profile_from_product(pkg, p)
hold = []  # EKPNP_SPREAD_HOLD=<GB>: a dummy allocation of that size made before every context, so that the next arena lands elsewhere
out = []
import torch
for k in range(n):
    if os.environ.get("EKPNP_SPREAD_HOLD"):
        hold.append(torch.empty(int(float(os.environ["EKPNP_SPREAD_HOLD"]) * 2**30), dtype=torch.uint8, device="cuda"))
    s = pkg.Solver(p)
    bench.product_pb_state(s, p, prof)
    s.fast_Poisson(); s.init_equilibrium(); s.step(8); s.synchronize()
    s.kernel_timing(True)
    t0 = time.perf_counter(); s.step(25); s.synchronize(); dt = time.perf_counter() - t0
    _, ms, _ = s.kernel_timing_get()
    ptrs = [s.field_device_ptr(f) for f in ("rho", "Ex", "T")]
    print(f"context {k}: {dt / 25 * 1e3:8.3f} ms/step, bulk {ms / 25:8.3f} ms, {grid[0] * grid[1] * grid[2] * 25 / dt / 1e6:8.1f} MLUPS, rho @ {ptrs[0]:#x}, placement {s.placement_report()}", flush=True)
    out.append(dt / 25 * 1e3)
    s.close()
a = np.array(out)
print(f"ms/step over {n} contexts: min {a.min():.3f} max {a.max():.3f} spread {(a.max() / a.min() - 1) * 100:.2f} %")
