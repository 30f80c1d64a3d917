#!/usr/bin/env python3
"""A diagnostic, not a test: N real RCCL ranks on the one GPU of the box (tests/_rccl_worker.py: ncclCommInitRank, halo ring,
EDGE gather, PHI exchange, all-reduce - the library's multi-GPU code path, RCCL wired through its socket transport) against the
CPU ORACLE, with a parent process that never touches the GPU - so that N may go up to 5: a box allows six processes on its card and
the launcher (torch.distributed.run) counts as one (six ranks were tried once: the box's process guard ended the run).  The suite's
own RCCL tests stop at 4 ranks: their parent holds a GPU context too.

    python tests/diagnostics/rccl_ranks_vs_oracle.py N [NXxNYxNZ] [out.json]

The transport's knobs come from the environment and reach every rank (EKPNP_EDGE_P2P=1, EKPNP_EDGE_CHUNKS=2, EKPNP_MERGED_FACES=0,
EKPNP_SLAB_IN_PLACE=1 ...).  Six steps from the perturbed start of the suite; exit code 1 if a field group misses the suite's
tolerance against the oracle (1e-9, velocities 1e-7) or the ranks' combined diagnostics differ from the oracle's.
Test infrastructure: the oracle is the checker here, as in tests/."""
import json
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402


def main():
    nprocs = int(sys.argv[1])
    shape = tuple(int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "40x6x37").lower().split("x"))
    out_path = sys.argv[3] if len(sys.argv) > 3 else None
    if not 1 <= nprocs <= 5:
        raise SystemExit("1 to 5 ranks (a box allows six processes on its card, and the launcher is one of them)")
    O = G.load_oracle()
    po = O.default_params(*shape)
    po.pb_iterations = 12  # (what the worker's initialization() runs; the 7 start fields below replace its result anyway)
    orc = O.Oracle(po)
    orc.initialization()
    st = O.perturb_fields(po, orc.fields())
    orc.set_fields(st)
    orc.fast_poisson()
    orc.init_equilibrium()
    orc.step(6)
    want, cur, um = orc.fields(), orc.current(), orc.umax()
    orc.close()
    with tempfile.TemporaryDirectory() as tmp:
        np.savez(os.path.join(tmp, "start.npz"), **st)
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, EKPNP_SLAB_OUT=tmp, EKPNP_SLAB_GRID="x".join(map(str, shape)), EKPNP_RCCL_FIELDS_ONLY="1", OMP_NUM_THREADS="1")
        env.setdefault("EKPNP_SLAB_IN_PLACE", "0")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nprocs}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "tests", "_rccl_worker.py")]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
        if r.returncode != 0:
            print(r.stderr[-4000:], file=sys.stderr)
            raise SystemExit(1)
        parts = sorted((dict(np.load(os.path.join(tmp, f"rank{k}.npz"))) for k in range(nprocs)), key=lambda d: int(d["z0"]))
    got = {k: np.concatenate([d[k] for d in parts], axis=0) for k in O.FIELDS}
    err = {k: float(v) for k, v in O.rel_l2(got, want).items()}
    ok = all(v <= (1e-7 if k == "u" else 1e-9) for k, v in err.items())
    diag_ok = all(abs(float(d["current"]) - cur) <= 1e-8 * abs(cur) and abs(float(d["umax"]) - um) <= 1e-6 * abs(um) + 1e-30 for d in parts)
    knobs = {k: v for k, v in sorted(os.environ.items()) if k.startswith("EKPNP_")}
    rec = {"ranks": nprocs, "grid": list(shape), "planes_per_rank": [int(d["rho"].shape[0]) for d in parts], "steps": 6, "knobs_from_environment": knobs,
           "rel_l2_vs_oracle": err, "within_tolerance": ok, "diagnostics_equal_the_oracles": diag_ok,
           "ranks_on_device": [int(d["ranks_on_device"]) for d in parts], "own_plane_transforms": [bool(d["own_passes"]) for d in parts]}
    print(json.dumps(rec))
    if out_path:
        json.dump(rec, open(out_path, "w"), indent=1)
    raise SystemExit(0 if ok and diag_ok else 1)


if __name__ == "__main__":
    main()
