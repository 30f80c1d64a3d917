"""At BASELINE's full cfg3 size (512^3, four lattices, 247 GB): do the cache-aware orders change a bit?

Two runs from bench.py's own start state (product Poisson-Boltzmann profile + closed-form 3-D perturbation), N steps each:
  A  plane order of the sweep, whole passes of the solve   (ekpnp_tune bulk_yband = 0, poisson_blocks = 1: rounds 1 - 4)
  B  the defaults at HEAD                                   (bands of 128 rows, three column blocks)
sha256 of every field's bytes after the run, both runs; they must be equal field by field.  The fields are hashed one at a
time (1 GiB each), never all on the host at once.  Usage: full_size_order_check.py [steps] [out.json] [grid]"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402
import bench  # noqa: E402

FIELDS = ("rho", "ux", "uy", "uz", "c", "cn", "phi", "Ex", "Ey", "Ez", "T")


def run(pkg, p, prof, steps, knobs):
    with pkg.Solver(p) as s:
        for k, v in knobs.items():
            s.tune(k, v)
        order = s.pass_order()
        bench.product_pb_state(s, p, prof)
        bench.apply_perturbation(s, None, p)
        s.fast_Poisson()
        s.init_equilibrium()
        s.synchronize()
        t = time.perf_counter()
        s.step(steps)
        s.synchronize()
        ms = (time.perf_counter() - t) / steps * 1e3
        out = {}
        for k in FIELDS:
            a = s.get_field(k)
            out[k] = {"sha256": hashlib.sha256(a.tobytes()).hexdigest(), "finite": bool(np.isfinite(a).all()), "absmax": float(np.abs(a).max())}
            del a
    return order, ms, out


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    out_path = sys.argv[2] if len(sys.argv) > 2 else None
    grid = tuple(int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "512x512x512").split("x"))
    pkg = G.load_package()
    p = pkg.default_params(*grid)
    prof, note = bench.pb_profile_from_product(pkg, p)
    res = {"grid": list(grid), "steps": steps, "start": note + " + closed-form 3-D perturbation", "runs": []}
    for label, knobs in (("plane order, whole passes", {"bulk_yband": 0, "poisson_blocks": 1}), ("defaults at HEAD", {})):
        order, ms, h = run(pkg, p, prof, steps, knobs)
        res["runs"].append({"label": label, "pass_order": order, "ms_per_step_wall": round(ms, 3), "fields": h})
        print(label, order, f"{ms:.3f} ms per step", flush=True)
    a, b = res["runs"][0]["fields"], res["runs"][1]["fields"]
    res["same_bits"] = {k: a[k]["sha256"] == b[k]["sha256"] for k in FIELDS}
    res["all_finite"] = all(v["finite"] for r in res["runs"] for v in r["fields"].values())
    res["ok"] = all(res["same_bits"].values()) and res["all_finite"] and res["runs"][0]["pass_order"] != res["runs"][1]["pass_order"]
    print(json.dumps({k: res[k] for k in ("same_bits", "all_finite", "ok")}), flush=True)
    if out_path:
        json.dump(res, open(out_path, "w"), indent=1)
    sys.exit(0 if res["ok"] else 1)


if __name__ == "__main__":
    main()
