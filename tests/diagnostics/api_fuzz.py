#!/usr/bin/env python3
"""A stateful bug hunt, not a test: random SEQUENCES of the C ABI's calls on small lattices, mirrored call by call on the CPU
oracle, fields compared whenever the sequence looks at them.

    python tests/diagnostics/api_fuzz.py FIRST_SEED COUNT [out.json]

What the straight-line parity tests do not reach is the library's bookkeeping between calls: is E still the central
difference of phi (lazy E), is the right-hand side the collide left still the one fast_Poisson may use, does a captured
graph still describe the step, may an intermediate step of a batch skip its moment stores, which population buffer is
current, is the state streamed or post-collision.  Every operation below flips some of those flags:

  step(n) | the split pair stream_collide_save + fast_Poisson | fast_Poisson alone (twice in a row too) | get_field of a random
  field | set_field of E / phi / c, cn / rho, u mid-run (the reference's arrays are plain device memory: main.cu may write
  them between calls) | init_equilibrium mid-run (main.cu:174 after a restart) | ekpnp_tune of a random knob | exposing an
  array through ekpnp_field_device_ptr | binding an array to caller memory (ekpnp_bind_field) and writing c / cn there on the
  device | invalidate_rhs | a full checkpoint into a NEW context of the other population mode
  or into a group of slabs (and back) | current / umax

The oracle does what the reference would do with the same calls (its arrays are the reference's arrays).  Exit code 1 if a
comparison misses the suite's tolerance or a call fails.  Test infrastructure: the oracle is the checker, as in tests/."""
import importlib.util
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as G  # noqa: E402

SOLVER_KNOBS = (("tri_partition", (0, 1, 2)), ("batch_moments", (0, 1)), ("lazy_efield", (0, 1)), ("merged_walls", (0, 1)), ("tri_wide", (0, 1)), ("bulk_yband", (-1, 0, 64)), ("poisson_blocks", (0, 1, 3)), ("poisson_zchunk", (0, 4)), ("ab_zchunk", (0, 1, 3)))
GROUP_KNOBS = (("bulk_yband", (-1, 0, 64)), ("edge_chunks", (1, 2, 3)), ("merged_faces", (0, 1)), ("lead_planes", (0, 1, 2)), ("batch_moments", (0, 1)), ("tri_partition", (0, 1, 2)), ("lazy_efield", (0, 1)))
TOL, TOL_U = 1e-9, 1e-7


def draw_params(O, rng):
    nx = int(rng.choice([rng.integers(1, 40), 64, rng.integers(65, 140)]))
    ny = int(rng.integers(1, 9))
    nz = int(rng.choice([rng.integers(5, 30), rng.integers(30, 66), 66, 67, rng.integers(68, 100)]))
    while nx * ny * nz > 60_000:
        ny = max(1, ny // 2)
        if ny == 1:
            nx = max(1, nx // 2)
    po = O.default_params(nx, ny, nz)
    po.pb_iterations = int(rng.integers(2, 12))
    po.n_lattices = int(rng.choice([3, 4, 4, 4]))
    if po.n_lattices < 4:
        po.Ra = 0.0
    if rng.random() < 0.5:
        po.uw, po.exf = 4e-4, 1.5e7
    if rng.random() < 0.5:
        po.voltage, po.voltage2 = -3.1e-3, -6.9e-3
    po.in_place = int(rng.random() < 0.4)
    return po


class Run:
    """the product side: one Solver or one Group, replaceable through a checkpoint"""

    def __init__(self, pkg, tp, po, nslabs):
        self.pkg, self.tp, self.po = pkg, tp, po
        self.nslabs = nslabs
        self.bound = {}  # caller-owned device arrays (torch tensors) the context is bound to: they outlive nothing but this context
        self.h = self._make(po, nslabs)

    def _make(self, po, nslabs):
        p = self.tp._mirror(self.pkg, po)
        return self.pkg.Solver(p) if nslabs == 1 else self.pkg.Group(p, nslabs, devices=[0] * nslabs)

    def close(self):
        self.h.close()


def one_sequence(pkg, O, tp, seed, log):
    rng = np.random.default_rng(seed)
    po = draw_params(O, rng)
    nl = po.n_lattices
    groups = {k: v for k, v in O.GROUPS.items() if not (nl == 3 and k == "T")}
    max_slabs = max(1, min(4, po.nz // 4))
    nslabs = 1 if rng.random() < 0.6 else int(rng.integers(2, max_slabs + 1)) if max_slabs >= 2 else 1
    ops = [f"{po.nx}x{po.ny}x{po.nz} nl={nl} ip={po.in_place} slabs={nslabs} uw={po.uw:g} v=({po.voltage:g},{po.voltage2:g})"]
    orc = O.Oracle(po)
    run = Run(pkg, tp, po, nslabs)
    tmp = tempfile.mkdtemp(prefix="ekpnp_fuzz_")
    worst = {}

    def compare(where):
        got, want = run.h.fields(), orc.fields()
        err = O.rel_l2(got, want, groups)
        for k, v in err.items():
            worst[k] = max(worst.get(k, 0.0), v)
        bad = {k: v for k, v in err.items() if not (v <= (TOL_U if k == "u" else TOL))}
        if bad:
            raise AssertionError(f"{where}: {bad}")

    try:
        orc.initialization()
        start = O.perturb_fields(po, orc.fields())
        orc.set_fields(start)
        run.h.set_fields(start)
        orc.fast_poisson()
        run.h.fast_Poisson()
        orc.init_equilibrium()
        run.h.init_equilibrium()
        steps_done = 0
        for _ in range(int(rng.integers(6, 16))):
            op = rng.choice(["step", "step", "step", "split", "poisson", "get", "set_E", "set_phi", "set_c", "set_mom", "reinit", "tune", "expose", "inval",
                             "checkpoint", "diag", "compare", "bind", "dev_write"])
            if op == "step":
                n = int(rng.integers(1, 6))
                ops.append(f"step({n})")
                orc.step(n)
                run.h.step(n)
                steps_done += n
            elif op == "split":
                ops.append("stream_collide_save; fast_Poisson")
                orc.stream_collide_save()
                orc.fast_poisson()
                run.h.stream_collide_save(0.0)
                run.h.fast_Poisson()
                steps_done += 1
            elif op == "poisson":
                k = int(rng.integers(1, 3))
                ops.append(f"fast_Poisson x{k}")
                for _i in range(k):
                    orc.fast_poisson()
                    run.h.fast_Poisson()
            elif op == "get":
                name = str(rng.choice(O.FIELDS))
                ops.append(f"get_field({name})")
                got, want = run.h.get_field(name), orc.field(name)
                scale = max(float(np.abs(want).max()), 1e-300)
                tol = 1e-6 if name in ("ux", "uy", "uz") else 1e-9
                if name in ("Ex", "Ey") or (name == "T" and nl == 3):
                    continue  # rounding noise in x-y-smooth runs: compared as a group in `compare`
                if not float(np.abs(got - want).max()) <= tol * scale + (1e-11 if name.startswith("u") else 0.0):
                    raise AssertionError(f"get_field({name}): max abs diff {float(np.abs(got - want).max()):.3e} of {scale:.3e}")
            elif op == "set_E":
                name = str(rng.choice(["Ex", "Ey", "Ez"]))
                f = 1.0 + 0.02 * float(rng.random())
                ops.append(f"set_field({name} *= {f:.4f})")
                v = orc.field(name).copy() * f + (1e3 if name != "Ez" else 0.0)
                orc.set_fields({name: v})
                run.h.set_field(name, v)
            elif op == "set_phi":
                f = 1.0 + 0.01 * float(rng.random())
                ops.append(f"set_field(phi *= {f:.4f})")
                v = orc.field("phi").copy() * f
                orc.set_fields({"phi": v})
                run.h.set_field("phi", v)
            elif op == "set_c":
                name = str(rng.choice(["c", "cn"]))
                f = 1.0 + 0.01 * float(rng.random())
                ops.append(f"set_field({name} *= {f:.4f})")
                v = orc.field(name).copy() * f
                orc.set_fields({name: v})
                run.h.set_field(name, v)
            elif op == "set_mom":
                name = str(rng.choice(["rho", "ux", "uz", "T"]))
                ops.append(f"set_field({name} *= 1.001)")
                v = orc.field(name).copy() * 1.001
                orc.set_fields({name: v})
                run.h.set_field(name, v)
            elif op == "reinit":
                ops.append("init_equilibrium")
                orc.init_equilibrium()
                run.h.init_equilibrium()
            elif op == "tune":
                name, vals = (SOLVER_KNOBS if run.nslabs == 1 else GROUP_KNOBS)[int(rng.integers(0, len(SOLVER_KNOBS if run.nslabs == 1 else GROUP_KNOBS)))]
                v = int(rng.choice(vals))
                ops.append(f"tune({name}, {v})")
                run.h.tune(name, v)
            elif op == "expose":
                if run.nslabs != 1:
                    continue
                name = str(rng.choice(["Ez", "phi", "rho", "c", "Ex"]))
                ops.append(f"field_device_ptr({name})")
                run.h.field_device_ptr(name)
            elif op == "bind":
                if run.nslabs != 1:
                    continue
                import torch

                name = str(rng.choice(["c", "cn", "rho", "Ez", "phi", "T", "ux"]))
                ops.append(f"bind_field({name}) to caller memory")
                t = torch.from_numpy(run.h.get_field(name)).to("cuda")  # main.cu's own allocation, holding what the library's array held
                run.bound[name] = t
                run.h.bind_field(name, t.data_ptr())
            elif op == "dev_write":
                # main.cu writes its OWN c / cn array on the device between two calls (ekpnp_bind_field: read at call time)
                cands = [k for k in ("c", "cn") if k in run.bound]
                if run.nslabs != 1 or not cands:
                    continue
                import torch

                name = str(rng.choice(cands))
                f = 1.0 + 0.01 * float(rng.random())
                ops.append(f"device write: bound {name} *= {f:.4f}")
                run.bound[name].mul_(f)
                torch.cuda.synchronize()
                orc.set_fields({name: run.bound[name].cpu().numpy()})
            elif op == "inval":
                if run.nslabs != 1:
                    continue
                ops.append("invalidate_rhs")
                run.h.invalidate_rhs()
            elif op == "checkpoint":
                path = os.path.join(tmp, "ck.bin")
                run.h.save_checkpoint(path)
                po2 = po.copy()
                po2.in_place = int(rng.random() < 0.5)
                ns2 = 1 if rng.random() < 0.5 else (int(rng.integers(2, max_slabs + 1)) if max_slabs >= 2 else 1)
                ops.append(f"checkpoint -> new {'context' if ns2 == 1 else f'group of {ns2}'} in_place={po2.in_place}")
                run.close()
                run.bound = {}
                run.nslabs = ns2
                run.h = run._make(po2, ns2)
                run.h.load_checkpoint(path)
            elif op == "diag":
                ops.append("current, umax")
                c1, c0 = run.h.current(), orc.current()
                u1, u0 = run.h.umax(), orc.umax()
                if not abs(c1 - c0) <= 1e-6 * abs(c0) + 1e-24:
                    raise AssertionError(f"current {c1!r} vs {c0!r}")
                if not abs(u1 - u0) <= 1e-5 * abs(u0) + 1e-12:
                    raise AssertionError(f"umax {u1!r} vs {u0!r}")
            else:
                ops.append("compare")
                compare(f"after {len(ops) - 1} operations")
        ops.append("compare (end)")
        compare("end")
        return True, ops, worst, None
    except Exception as e:  # noqa: BLE001
        return False, ops, worst, f"{type(e).__name__}: {e}"
    finally:
        try:
            run.close()
        except Exception:  # noqa: BLE001
            pass
        orc.close()
        for f in os.listdir(tmp):
            os.remove(os.path.join(tmp, f))
        os.rmdir(tmp)


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    out_path = sys.argv[3] if len(sys.argv) > 3 else None
    pkg, O = G.load_package(), G.load_oracle()
    spec = importlib.util.spec_from_file_location("tp", os.path.join(ROOT, "tests", "test_parity_gpu.py"))
    tp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tp)
    failures, worst_all, n_ops = [], {}, 0
    t0 = time.time()
    for seed in range(first, first + count):
        ok, ops, worst, err = one_sequence(pkg, O, tp, seed, None)
        n_ops += len(ops) - 1
        for k, v in worst.items():
            worst_all[k] = max(worst_all.get(k, 0.0), v)
        print(("ok  " if ok else "BAD ") + f"{seed}: " + " | ".join(ops) + ("" if ok else f"  => {err}"), flush=True)
        if not ok:
            failures.append({"seed": seed, "ops": ops, "error": err})
    summary = {"first_seed": first, "count": count, "operations": n_ops, "failed": len(failures), "seconds": round(time.time() - t0, 1), "worst_rel_l2": worst_all}
    print(json.dumps(summary))
    if out_path:
        json.dump({"summary": summary, "failures": failures}, open(out_path, "w"), indent=1)
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
