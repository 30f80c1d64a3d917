"""time_vs_reference.py NXxNYxNZ [steps] - the full step of this library on one grid, for the comparison with the
reference's OWN kernels on the same GPU (oracle/_ref/ref_driver_<grid> time <steps>, built by
oracle/build_ref.sh <grid>).  Same physics (reference defaults), same start (initialization + init_equilibrium)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as G
pkg = G.load_package()
nx, ny, nz = (int(v) for v in sys.argv[1].split("x"))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
p = pkg.default_params(nx, ny, nz)
with pkg.Solver(p) as s:
    s.call("init_fields"); s.init_equilibrium(); s.step(5); s.synchronize()   # like ref_driver time0: uniform fields
    t0 = time.perf_counter(); s.step(steps); s.synchronize(); dt = time.perf_counter() - t0
    print(f"ekpnp step on this GPU: {steps} steps of {nx}x{ny}x{nz} in {dt:.4f} s = {1e3 * dt / steps:.3f} ms/step = {steps * nx * ny * nz / dt / 1e6:.2f} MLUPS", flush=True)
