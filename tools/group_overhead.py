"""group_overhead.py [NXxNYxNZ] [nslabs] [steps] - what does driving P slabs from ONE host thread cost at short steps?

ekpnp_group_* (ekpnp_main --gpus N) issues every slab's launches from one thread, about 25 per slab and step, with a
hipSetDevice hop before each slab.  On cfg4 over 8 GPUs a slab's step is only ~10 ms long.  Measured here on the one GPU
of the box: the lattice as `nslabs` in-place slabs on device 0 (transport: device copies; same events, same call order as
over 8 devices) against the SAME lattice in one in-place context:
  * host_enqueue_ms_per_step: wall time of ekpnp_group_step(n) / n WITHOUT waiting for the device - the host's share;
    as long as it is below the device's step time the host runs ahead and costs nothing;
  * ms_per_step of both shapes, device-synchronised - the difference is the slab machinery on the device (boundary-plane
    launches, pack / unpack, 2 x 75 MB halo copies per slab, the distributed z solve) plus whatever the host could not hide.
Default lattice 512x512x768: 8 slabs of 96 planes fit one MI355X in place next to each other (512x512x1024 does not)."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import __graft_entry__ as G  # noqa: E402


def start(run, p, shape_of):
    nz = p.nz
    z = np.arange(nz, dtype=np.float64)
    lam = np.sqrt(p.eps * p.kB * p.roomT / p.electron / (2 * p.chargeinf * p.convertCtoCharge))
    vt = p.kB * p.roomT / p.electron
    wall = lambda zeta, d: 4 * vt * np.arctanh(np.tanh(zeta / (4 * vt)) * np.exp(-d / lam))  # noqa: E731
    phi = wall(p.voltage, z * p.dz) + wall(p.voltage2, (nz - 1 - z) * p.dz)
    col = lambda v: np.broadcast_to(np.asarray(v)[:, None, None], shape_of)  # noqa: E731
    run.set_field("phi", col(phi))
    run.set_field("c", col(p.chargeinf * np.exp(-phi / vt)))
    run.set_field("cn", col(p.chargeinf * np.exp(phi / vt)))
    run.set_field("rho", col(np.full(nz, p.rho0)))
    run.set_field("T", col(p.TH * (p.Lz - p.dz * z) / p.Lz))
    import bench

    if not hasattr(run, "z0"):
        run.z0 = 0  # a group speaks whole-lattice arrays
    bench.apply_perturbation(run, None, p)  # x-y-z dependent start: the two shapes are compared field by field below
    run.fast_Poisson()
    run.init_equilibrium()


def measure(run, steps):
    run.step(3)
    run.synchronize()
    t0 = time.perf_counter()
    run.step(steps)
    t_enq = time.perf_counter() - t0
    run.synchronize()
    t_all = time.perf_counter() - t0
    return t_enq / steps * 1e3, t_all / steps * 1e3


def main():
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    grid = pos[0] if len(pos) > 0 else "512x512x768"
    nslabs = int(pos[1]) if len(pos) > 1 else 8
    steps = int(pos[2]) if len(pos) > 2 else 20
    nx, ny, nz = (int(v) for v in grid.split("x"))
    pkg = G.load_package()
    p = pkg.default_params(nx, ny, nz)
    p.in_place = 1
    out = {"grid": [nx, ny, nz], "nslabs": nslabs, "steps": steps}
    if "--group-only" in sys.argv:  # under rocprofv3 --kernel-trace: tools/gpu_busy_from_trace.py reads the device's idle time off the trace
        with pkg.Group(p, nslabs, devices=[0] * nslabs) as g:
            start(g, p, g.shape)
            enq, ms = measure(g, steps)
            print(json.dumps({"group": {"host_enqueue_ms_per_step": round(enq, 3), "ms_per_step": round(ms, 3)}}))
        return
    with pkg.Solver(p) as s:
        start(s, p, s.shape)
        enq, ms = measure(s, steps)
        ref = {k: s.get_field(k) for k in ("rho", "c", "phi", "ux", "uz", "T")}
        out["one_context"] = {"host_enqueue_ms_per_step": round(enq, 3), "ms_per_step": round(ms, 3), "MLUPS": round(nx * ny * nz / ms / 1e3, 1)}
    with pkg.Group(p, nslabs, devices=[0] * nslabs) as g:
        start(g, p, g.shape)
        enq, ms = measure(g, steps)
        # the decomposition at full plane width against the single context, field by field (same start, same number of steps)
        err = {}
        for k, want in ref.items():
            got = g.get_field(k)
            err[k] = float(np.sqrt(((got - want) ** 2).sum() / (want**2).sum()))
            del got
        out["group_vs_one_context_rel_l2"] = err
        out["group"] = {"host_enqueue_ms_per_step": round(enq, 3), "ms_per_step": round(ms, 3), "MLUPS": round(nx * ny * nz / ms / 1e3, 1),
                        "device_ms_per_slab_step": round(ms / nslabs, 3)}
    out["group_over_one_context"] = round(out["group"]["ms_per_step"] / out["one_context"]["ms_per_step"], 4)
    # what the host would need per step if every slab had its own GPU: the slabs then run side by side, so the
    # enqueue time of ALL slabs must stay below ONE slab's device time
    out["host_bound_if_8_gpus"] = out["group"]["host_enqueue_ms_per_step"] > out["group"]["device_ms_per_slab_step"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
