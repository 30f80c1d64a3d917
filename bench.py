#!/usr/bin/env python3
"""bench.py — MLUPS of the full EK-PNP step (stream_collide_save + fast_Poisson, main.cu:189-200).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2|cfg1|NXxNYxNZ]

N=1: cfg3 of BASELINE.json (512x512x512, four D3Q27 lattices + Poisson) when it fits the GPU,
otherwise cfg2 (256^3, f+h+hn).  N>1 (one rank per GPU: launched by torch.distributed.run, or by
bench.py itself when invoked plainly - the ranks are then child processes) runs BASELINE.json's
multi-GPU configurations BY NAME: N = 2..7 -> cfg4 (512x512x1024 split into N z slabs: 512 planes
per rank at N=2 = the weak-scaling partner of cfg3@1, 256 at N=4 = strong scaling), N >= 8 -> cfg5
(1024^3, 128 planes per rank at N=8: 134 M nodes per rank like cfg3@1, weak scaling);
`--workload cfg4` is accepted at any N (128 planes per rank at N=8, the strong-scaling end),
`--workload cfg3 --weak` keeps the 512^3-slab-per-rank channel of the earlier rounds, `--scale-z D`
divides the z extent for rehearsals on a one-GPU box (labelled in the line).  z-slab decomposition,
halo exchange over RCCL inside libekpnp.so (ekpnp_slab_attach_comm); torch.distributed (gloo) is
the control plane only: rendezvous, barrier, max over ranks.

One JSON line on rank 0.  `value` counts lattice-node updates of all ranks per wall second of
the timed region (inputs resident in HBM, barrier + device sync on both sides, max over ranks).
`roofline` is for the dominant kernel (the fused pull-stream/collide bulk kernel): algorithmic
bytes per launch / its mean launch duration measured with HIP events on the kernel's stream.
`cpu_baseline` is the CPU oracle (test infrastructure, never the measured product) timed on the
host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# CPUs this process may run on, read BEFORE anything brings an OpenMP runtime in: with OMP_PROC_BIND
# set (the CPU-baseline child below) the runtime binds the initial thread to its first place when it
# starts, and the affinity mask read afterwards is that one core
try:
    _CPUS_AT_START = len(os.sched_getaffinity(0))
except AttributeError:
    _CPUS_AT_START = os.cpu_count() or 1

# dmabuf IPC: RCCL across processes needs it on this pool (hipIpcGetMemHandle fails without it).  Set HERE, before torch
# (and with it the HSA runtime) is imported, so that ranks started by `python -m torch.distributed.run ... bench.py` - the
# driver's launch - have it exactly like the ranks bench.py spawns itself (spawn_ranks); an explicit setting wins.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def b_alg_lbm(nl: int) -> int:
    """Algorithmic bytes per node of one bulk-kernel launch (LBM part):
    every population read once and written once (nl*27*16), E read (24, nl>1), macroscopic
    writes rho,u (32) + c,cn (16, nl>1) + T (8, nl>3)."""
    return nl * 27 * 16 + (24 + 16 if nl > 1 else 0) + 32 + (8 if nl > 3 else 0)


def b_alg_step(nl: int) -> int:
    """Algorithmic bytes per node of the full step: LBM part + Poisson compulsory I/O 48
    (read c,cn; write phi, E).  cfg3: 1856, cfg2: 1416, cfg1: 464 (no Poisson I/O credited)."""
    return b_alg_lbm(nl) + (48 if nl > 1 else 0)


def parse_workload(name: str, free_bytes: int, in_place: bool = False):
    """-> (name, grid, lattices, in_place).  auto: cfg3 with two population buffers if 248 GB are
    free, else cfg3 in place (146 GB), else cfg2."""
    if name == "auto":
        fields = 16 * 512**3 * 8
        need3 = 2 * 4 * 27 * 514 * 512 * 512 * 8 + fields
        need3_ip = 4 * 27 * (514 + 65) * 512 * 512 * 8 + fields
        if in_place or free_bytes <= need3 * 1.02:
            in_place = True
        name = "cfg3" if free_bytes > (need3_ip if in_place else need3) * 1.02 else "cfg2"
        if name == "cfg2":
            in_place = False
    if name == "cfg3":
        return name, (512, 512, 512), 4, in_place
    if name == "cfg2":
        return name, (256, 256, 256), 3, in_place
    if name == "cfg1":
        return name, (64, 64, 64), 1, in_place
    nx, ny, nz = (int(v) for v in name.lower().split("x"))
    return name, (nx, ny, nz), 4, in_place


CONFIGS = {  # BASELINE.json configs[0..4]: global grid, lattices
    "cfg1": ((64, 64, 64), 1),
    "cfg2": ((256, 256, 256), 3),
    "cfg3": ((512, 512, 512), 4),
    "cfg4": ((512, 512, 1024), 4),
    "cfg5": ((1024, 1024, 1024), 4),
}
NODES_PER_RANK_AT_1 = 512**3  # cfg3, the N=1 line every N>1 line is compared with


def device_need_bytes(nx: int, ny: int, nzl: int, nl: int, in_place: bool) -> int:
    """device memory of one context / slab of nzl planes (csrc/capi.hip create_impl): tiled populations with two ghost
    planes (+ the in-place shift), 11 fields + rhs + half spectrum"""
    pplane = ((nx + 63) // 64) * 27 * 64 * ny * 8
    shift = (min(max(nzl // 4, 1), 64) + 1) if in_place else 0
    pops = (1 if in_place else 2) * nl * (nzl + 2 + shift) * pplane
    return pops + 16 * nx * ny * nzl * 8


def select_workload(name: str, world: int, free_bytes: int, in_place: bool = False, weak: bool = False, scale_z: int = 1) -> dict:
    """Which lattice `bench.py --gpus world --workload name` runs, and what its line may call itself.

    auto: N=1 -> cfg3 (cfg2 if it does not fit), N=2..7 -> cfg4, N>=8 -> cfg5.  Named grids are GLOBAL and split into
    `world` z slabs; --weak makes the named grid ONE RANK's slab (the channel is `world` of them on top of each other).
    `scaling` is "weak" when a rank owns as many nodes as the N=1 line's cfg3 (or with --weak), else "strong"."""
    if world < 1 or world > 16:
        raise ValueError("1 to 16 ranks (at most 16 z slabs)")
    if scale_z < 1:
        raise ValueError("--scale-z must be >= 1")
    if name == "auto" and world > 1:
        name = "cfg5" if world >= 8 else "cfg4"
    if world == 1 and not weak and scale_z == 1:
        wname, grid, nl, in_place = parse_workload(name, free_bytes, in_place)
    else:
        wname = "cfg3" if name == "auto" else name
        if wname in CONFIGS:
            grid, nl = CONFIGS[wname]
        else:
            grid, nl = tuple(int(v) for v in wname.lower().split("x")), 4
            if len(grid) != 3:
                raise ValueError(f"workload {wname!r}: cfg1..cfg5 or NXxNYxNZ")
    nx, ny, nz = grid
    if weak:
        nz *= world
    nz //= scale_z
    if world > 1 and nz // world < 4:
        raise ValueError(f"{nz} planes over {world} ranks: each z slab needs at least 4 planes")
    planes = [(r + 1) * nz // world - r * nz // world for r in range(world)]
    if world > 1 or weak or scale_z > 1:
        if not in_place and free_bytes <= device_need_bytes(nx, ny, max(planes), nl, False) * 1.02:
            in_place = True  # every rank takes the same decision: free_bytes is the minimum over the ranks
    nodes_per_rank = nx * ny * nz / world
    if world == 1:
        scaling, note = "weak", "N=1: the line every N>1 line is compared with"
    elif weak:
        scaling, note = "weak", f"--weak: every rank owns a {grid[0]}x{grid[1]}x{grid[2] // scale_z} slab, the channel grows with N"
    elif nodes_per_rank == NODES_PER_RANK_AT_1:
        scaling, note = "weak", "a rank owns as many nodes as the one GPU of the N=1 line (cfg3, 134 M)"
    else:
        scaling, note = "strong", f"fixed {nx}x{ny}x{nz} lattice split over the ranks: {nodes_per_rank / NODES_PER_RANK_AT_1:.3g} of the N=1 line's nodes per rank"
    label = f"{wname}: {nx}x{ny}x{nz} D3Q27 x{nl} lattices" + (" + spectral Poisson" if nl > 1 else "")
    if world > 1:
        pl = f"{planes[0]}" if min(planes) == max(planes) else f"{min(planes)}-{max(planes)}"
        label += f", z-slabs of {pl} planes over {world} GPUs"
    if scale_z > 1:
        label += f" - REHEARSAL: z extent divided by {scale_z}, not the BASELINE size"
    return {"name": wname, "grid": (nx, ny, nz), "lattices": nl, "in_place": bool(in_place), "scaling": scaling, "scaling_note": note,
            "nodes_per_rank": int(nodes_per_rank), "planes_per_rank": planes, "label": label, "rehearsal_scale_z": scale_z}


def comm_block(raw, steps: int, halo_bytes_formula: int, per_rank_wait=None) -> dict:
    """The `comm` object of an N>1 (or --force-slab) line from ekpnp_comm_timing_get's sums on rank 0: per exchange kind
    the bytes a rank sends per step, the time its COMPUTE stream waited for the exchange and the time the exchange took
    on the comm stream (both HIP events, ms per step); per_rank_wait = total wait per step of every rank."""
    steps = max(1, steps)
    out = {"source": "HIP events around the exchanges inside libekpnp.so (ekpnp_comm_timing_get), rank 0",
           "halo_bytes_per_step": int(raw["halo"]["bytes_sent"]) * (raw["halo"]["n"] // steps if raw["halo"]["n"] else 0),
           "halo_bytes_per_step_formula": int(halo_bytes_formula)}
    tw = tt = 0.0
    for k in ("halo", "edge", "phi"):
        r = raw[k]
        out[k] = {"exchanges_per_step": r["n"] / steps, "bytes_sent_per_exchange": int(r["bytes_sent"]),
                  "wait_ms_per_step": round(r["wait_ms"] / steps, 4), "transfer_ms_per_step": round(r["transfer_ms"] / steps, 4)}
        tw += r["wait_ms"] / steps
        tt += r["transfer_ms"] / steps
    out["wait_ms_per_step"] = round(tw, 4)
    out["transfer_ms_per_step"] = round(tt, 4)
    if per_rank_wait is not None:
        out["wait_ms_per_step_by_rank"] = [round(float(v), 4) for v in per_rank_wait]
    return out


COMM_AB_LEGS = (  # (label, [(knob, value) ...]) - what ONE GPU cannot decide; defaults in AB_DEFAULTS
    ("defaults", []),
    ("inline_exchanges=0", [("inline_exchanges", 0)]),
    ("comm_cus=8", [("comm_cus", 8)]),
    ("lead_planes=0", [("lead_planes", 0)]),
    ("edge_chunks=4", [("edge_chunks", 4)]),
    ("edge_chunks=4 comm_cus=8", [("edge_chunks", 4), ("comm_cus", 8)]),
    ("edge_p2p=1", [("edge_p2p", 1)]),  # the EDGE gather as direct send / receive pairs with every peer instead of ncclAllGather
)
AB_DEFAULTS = {"inline_exchanges": ("EKPNP_INLINE_EXCHANGES", 1), "comm_cus": ("EKPNP_COMM_CUS", 0), "lead_planes": ("EKPNP_SLAB_LEAD_PLANES", 2),
               "edge_chunks": ("EKPNP_EDGE_CHUNKS", 1), "edge_p2p": ("EKPNP_EDGE_P2P", 0)}


def ab_baseline() -> dict:
    """the knob values the timed region ran with: the library's defaults unless the environment says otherwise"""
    return {k: int(os.environ.get(env, str(dflt))) for k, (env, dflt) in AB_DEFAULTS.items()}


def phases_of(dt_s: float, steps: int, k_ms: float, poisson_ms: float, n_solves: int) -> dict:
    steps = max(1, steps)
    bulk, pois = k_ms / steps, poisson_ms / max(1, n_solves)
    return {"collide_bulk": round(bulk, 4), "poisson": round(pois, 4), "rest": round(dt_s / steps * 1e3 - bulk - pois, 4)}


def min_max_by_rank(dicts: list) -> dict:
    """[{key: value} per rank] -> {"min": {...}, "max": {...}} (the spread the slowest rank hides behind max-over-ranks timing)"""
    keys = list(dicts[0])
    return {"min": {k: round(min(d[k] for d in dicts), 4) for k in keys}, "max": {k: round(max(d[k] for d in dicts), 4) for k in keys}}


class HeadlineGuard:
    """Keeps the measured line safe from what runs AFTER the timed region (the knob A/Bs, the copy probe, the CPU baseline).

    Those legs are outside `value`, but they run before the ONE JSON line is printed, and on the first real multi-GPU run
    they are the least rehearsed code: an RCCL exchange under an untried knob that never completes, or an exception on one
    rank while its peers sit in a collective, would take the headline with it.  So every rank arms a timer when its timed
    region has closed; rank 0 hands it the line as it stands (update() after every finished leg).  If the legs are not done
    `deadline_s` later, rank 0 writes that line - marked `after_the_fact.status = "abandoned"`, naming the leg - straight to the
    saved stdout descriptor, and every rank ends itself with os._exit(0): the measurement is complete and valid, the process
    is not allowed to hang on its appendix.  (os._exit, not an exec and not a signal to anybody else: the driver's launcher
    sees N ranks that ended with 0.)  finish() prints the complete line instead and disarms; whichever comes first wins."""

    def __init__(self, rank: int, deadline_s: float, fd: int = 1, exit_fn=None):
        import threading

        self.rank, self.deadline_s, self.fd = rank, float(deadline_s), fd
        self._lock = threading.Lock()
        self._line = None       # rank 0: the line as JSON text, as of the last finished leg
        self._leg = "start"
        self._done = False
        self._timer = None
        self._exit = exit_fn if exit_fn is not None else os._exit
        self._threading = threading

    def arm(self, out=None):
        with self._lock:
            if out is not None:
                self._line = dict(out)
            if self.deadline_s > 0 and self._timer is None:
                self._timer = self._threading.Timer(self.deadline_s, self._fire)
                self._timer.daemon = True
                self._timer.start()

    def update(self, out=None, leg=None):
        with self._lock:
            if out is not None:
                self._line = dict(out)
            if leg is not None:
                self._leg = leg

    def _write(self, obj):
        data = (json.dumps(obj) + "\n").encode()
        while data:
            data = data[os.write(self.fd, data):]

    def _fire(self):
        with self._lock:
            if self._done:
                return
            self._done = True
            if self._line is not None:
                line = dict(self._line)
                line["after_the_fact"] = {"status": "abandoned", "leg": self._leg, "deadline_s": self.deadline_s,
                                          "note": "the timed region had closed and `value` stands; a leg that runs after it did not finish in time"}
                self._write(line)
            print(f"bench.py: rank {self.rank}: the after-the-fact leg '{self._leg}' did not finish within {self.deadline_s:.0f} s; "
                  "the headline was measured before it and is printed; leaving", file=sys.stderr, flush=True)
            self._exit(0)

    def finish(self, out=None):
        """the normal end: print the complete line (rank 0) and disarm.  False if the timer won the race (the line is out already)."""
        with self._lock:
            if self._done:
                return False
            self._done = True
            if self._timer is not None:
                self._timer.cancel()
            if out is not None:
                self._write(out)
            return True


def comm_ab_leg(label, knobs, sol, runner, steps, barrier, dist, torch, world, base=None):
    """ONE leg of the after-the-fact knob A/B: `steps` steps of the live context under `knobs` (set through ekpnp_tune on every
    rank, in the same order), timed like the headline (barrier + sync both sides, max over ranks), with what the compute
    stream waited for per exchange (max over ranks) and the solve's stage times (max over ranks).  Outside `value`.
    A knob that one rank cannot set (say, a CU mask its driver refuses) must not cost the headline line: the ranks agree on
    the outcome of the tune calls over the control plane BEFORE anybody steps, and a refused leg is reported as such."""
    def agree_failed(failed: bool) -> bool:
        if dist is None:
            return failed
        flag = torch.tensor([1 if failed else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        return bool(int(flag.item()))

    def restore():
        for k, _ in knobs:
            if base is not None and k in base:
                try:
                    sol.tune(k, base[k])
                except Exception:  # noqa: BLE001
                    pass

    err = None
    try:
        for k, v in knobs:
            sol.tune(k, v)
    except Exception as e:  # noqa: BLE001
        err = str(e)
    if agree_failed(err is not None):
        restore()
        return {"knob": label, "error": err or "refused on another rank"}
    runner.step(2)  # the first steps after a change re-make streams / first-touch the chunked buffers
    barrier()
    sol.kernel_timing(True)
    t0 = time.perf_counter()
    runner.step(steps)
    barrier()
    dt = time.perf_counter() - t0
    n_launch, k_ms, _ = sol.kernel_timing_get()
    _, stages = sol.poisson_stage_timing_get()
    n_solves, poisson_ms = sol.phase_timing_get()
    raw = sol.comm_timing_get()
    sol.kernel_timing(False)
    restore()  # back to what the timed region ran with
    vec = [dt * 1e3 / steps, k_ms / steps, poisson_ms / max(1, n_solves)] + [raw[k]["wait_ms"] / steps for k in ("halo", "edge", "phi")] \
        + [stages[k] / max(1, n_solves) for k in ("stage1", "edge_exchange", "stage2", "phi_exchange", "stage3")]
    if dist is not None:
        tv = torch.tensor(vec, dtype=torch.float64)
        dist.all_reduce(tv, op=dist.ReduceOp.MAX)
        vec = [float(v) for v in tv]
    names = ("ms_per_step", "collide_bulk_ms", "poisson_ms", "halo_wait_ms", "edge_wait_ms", "phi_wait_ms",
             "stage1_ms", "edge_exchange_ms", "stage2_ms", "phi_exchange_ms", "stage3_ms")
    out = {"knob": label, "steps": steps}
    out.update({n: round(v, 4) for n, v in zip(names, vec)})
    return out


def step_traffic_of(rec: dict):
    """HBM bytes of one steady-state step from a workload's record in profiles/pmc_traffic.json (tools/summarize_profile.py):
    the sum over the step's kernels of launches per step x counter bytes per launch, or None when the record has no such
    table or one of the step's kernels has no counter value."""
    st = (rec or {}).get("step")
    if not st or st.get("kernels_without_counters"):
        return None
    tot = 0.0
    for e in st.get("kernels", []):
        if e.get("hbm_bytes_per_launch") is None:
            return None
        tot += e["launches_per_step"] * e["hbm_bytes_per_launch"]
    return tot or None


def host_cpu_budget():
    """(cores the CPU baseline may use, how that was found).  A container's share of the host is its cgroup CPU quota
    (cpu.max: "<quota> <period>" or "max"), not the affinity mask - the GPU box shows every hardware thread of the
    host in the mask.  Without a quota: the PHYSICAL cores among the CPUs of the mask."""
    import math

    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            q, per = open(path).read().split()[:2]
            if q != "max" and float(per) > 0:
                return max(1, min(_CPUS_AT_START, math.ceil(float(q) / float(per)))), f"cgroup quota {path}: {q} {per}"
        except (OSError, ValueError):
            pass
    try:  # cgroup v1
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and per > 0:
            return max(1, min(_CPUS_AT_START, math.ceil(q / per))), f"cgroup v1 quota {q}/{per}"
    except (OSError, ValueError):
        pass
    try:
        mask = os.sched_getaffinity(0)
        cores = set()
        for cpu in mask:
            base = f"/sys/devices/system/cpu/cpu{cpu}/topology/"
            cores.add((open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip()))
        if cores:
            return len(cores), f"no cgroup CPU quota: physical cores among the {len(mask)} CPUs of the affinity mask"
    except (OSError, AttributeError):
        pass
    return max(1, _CPUS_AT_START), "no cgroup CPU quota, no topology: CPUs of the affinity mask"


def gouy_chapman_state(sol, p):
    """Closed-form Poisson-Boltzmann start for tall channels.  The reference's initialization()
    is a Picard iteration relaxed with PB_omega = 0.05 (LBM.cu:89-106); its lowest z mode is
    amplified by (Lz / (pi lambda_D))^2 per sweep, so it diverges once NZ exceeds ~180 planes at
    dz = 1e-8 (lambda_D = 9.2 dz) - cfg2/cfg3 cannot be started with it.  The two double layers
    do not overlap there, so the bench seeds each wall with the Gouy-Chapman solution
    tanh(e phi/4kT) = tanh(e zeta/4kT) exp(-z/lambda_D) and the Boltzmann concentrations
    (LBM.cu:144-145), then lets fast_Poisson make phi and E consistent.  Untimed."""
    nz, ny, nx = sol.shape
    z = (np.arange(nz) + sol.z0).astype(np.float64)
    lam = np.sqrt(p.eps * p.kB * p.roomT / p.electron / (2 * p.chargeinf * p.convertCtoCharge))
    vt = p.kB * p.roomT / p.electron
    def wall(zeta, dist):
        return 4 * vt * np.arctanh(np.tanh(zeta / (4 * vt)) * np.exp(-dist / lam))
    phi = wall(p.voltage, z * p.dz) + wall(p.voltage2, (p.nz - 1 - z) * p.dz)
    col = lambda v: np.broadcast_to(v[:, None, None], sol.shape)  # noqa: E731
    sol.set_field("phi", col(phi))
    sol.set_field("c", col(p.chargeinf * np.exp(-phi / vt)))
    sol.set_field("cn", col(p.chargeinf * np.exp(phi / vt)))
    sol.set_field("rho", col(np.full(nz, p.rho0)))
    sol.set_field("T", col(p.TH * (p.Lz - p.dz * z) / p.Lz))  # LBM.cu:127
    zero = np.zeros(sol.shape)
    for k in ("ux", "uy", "uz", "Ex", "Ey", "Ez"):
        sol.set_field(k, zero)


def apply_perturbation(sol, O, p):
    """initialization() state + closed-form 3-D perturbation."""
    f = {k: sol.get_field(k) for k in ("rho", "c", "cn", "T")}
    nz, ny, nx = sol.shape
    z = (np.arange(nz) + sol.z0)[:, None, None]
    y = np.arange(ny)[None, :, None]
    x = np.arange(nx)[None, None, :]
    X, Y, Z = 2 * np.pi * x / p.nx, 2 * np.pi * y / p.ny, np.pi * z / (p.nz - 1)
    sZ = np.sin(Z)
    sol.set_field("c", f["c"] * (1 + 0.02 * np.sin(X) * np.cos(2 * Y) * sZ))
    sol.set_field("cn", f["cn"] * (1 + 0.02 * np.cos(2 * X) * np.sin(Y) * sZ))
    sol.set_field("T", f["T"] + 0.05 * np.sin(X + Y) * sZ)
    sol.set_field("rho", f["rho"] * (1 + 1e-6 * np.cos(X) * np.cos(Y) * sZ))
    del f
    sol.set_field("ux", 1e-4 * sZ * np.sin(X) * np.cos(Y))
    sol.set_field("uy", -0.7e-4 * sZ * np.cos(X) * np.sin(2 * Y))
    sol.set_field("uz", 0.5e-4 * sZ**2 * np.cos(X) * np.cos(Y))


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _time_oracle(O, p, threads: int, budget_s: float, perturb: bool):
    """MLUPS of the oracle's full step on `threads` cores, about budget_s seconds of it: one untimed step, one timed step
    to size the sample, then k steps - or, when a single step already takes more than half the budget (one core on 128^3),
    that one timed step IS the sample."""
    o = O.Oracle(p)
    used = O.set_threads(threads)
    o.threads = used
    o.initialization()
    if perturb:
        o.set_fields(O.perturb_fields(p, o.fields()))
        o.fast_poisson()
    o.init_equilibrium()
    o.step(1)
    t0 = time.perf_counter()
    o.step(1)
    per = time.perf_counter() - t0
    k, dt = 1, per
    if per <= 0.5 * budget_s:
        k = max(2, min(500, int(budget_s / max(per, 1e-4))))
        t0 = time.perf_counter()
        o.step(k)
        dt = time.perf_counter() - t0
    n = o.n
    o.close()
    return n * k / dt / 1e6, k, dt, used


def cpu_baseline(nl: int, budget_s: float = 20.0):
    """BASELINE.md §3, to the letter.  The reference has no CPU path (every compute routine is a __global__ kernel), so the
    number beside the GPU line is the CPU oracle (kind "port": our restatement of the same step, test infrastructure) timed
    on this box's host cores, after the timed GPU region:
      * the multi-lattice point: the bench's physics (its lattices + Poisson) on 128^3, on all cores - this is `value` -
        and on ONE core (x-y uniform start and one timed step there: a step is seconds long, and timing is data-independent);
      * cfg1 of BASELINE.json (64x64x64, fluid lattice only, body-force channel) on all cores and on ONE core.
    MLUPS and the achieved-bandwidth estimate B_alg x LUPS for every leg.  Bounded: about budget_s seconds of timed steps."""
    O = G.load_oracle()
    # OMP_NUM_THREADS if set, else this container's CPU share: its cgroup quota, or the physical cores of its mask
    if os.environ.get("OMP_NUM_THREADS"):
        cores, cores_source = max(1, int(os.environ["OMP_NUM_THREADS"])), "OMP_NUM_THREADS"
    else:
        cores, cores_source = host_cpu_budget()
    nlat = nl if nl > 1 else 4  # (a fluid-only bench still reports the multi-lattice point BASELINE.md asks for)
    legs = []
    shape = (128, 128, 128)

    def multi(th, perturb, budget):
        p = O.default_params(*shape)
        p.pb_iterations = 3 if perturb else 1
        p.n_lattices = nlat
        if nlat < 4:
            p.Ra = 0.0
        v, k, dt, used = _time_oracle(O, p, th, budget, perturb)
        leg = {"workload": f"{shape[0]}x{shape[1]}x{shape[2]} D3Q27 x{nlat} lattices + Poisson", "cores": used, "value": round(v, 3), "steps": k,
               "seconds": round(dt, 2), "achieved_GBps": round(v * 1e6 * b_alg_step(nlat) / 1e9, 2)}
        legs.append(leg)
        return leg

    main_leg = multi(cores, True, 0.4 * budget_s)
    if cores > 1:
        multi(1, False, 0.1 * budget_s)
    p1 = O.default_params(64, 64, 64)
    p1.n_lattices, p1.chargeinf, p1.Ra, p1.TH, p1.exf, p1.pb_iterations = 1, 0.0, 0.0, 0.0, 1e9, 1
    for th in ((cores, 1) if cores > 1 else (1,)):
        v1, k1, dt1, used1 = _time_oracle(O, p1, th, 0.15 * budget_s, False)
        legs.append({"workload": "cfg1: 64x64x64 D3Q27 fluid lattice only (exf = 1e9, no ions)", "cores": used1, "value": round(v1, 3), "steps": k1,
                     "seconds": round(dt1, 2), "achieved_GBps": round(v1 * 1e6 * b_alg_step(1) / 1e9, 2)})
    return {
        "value": main_leg["value"],
        "unit": "MLUPS",
        "cores": main_leg["cores"],
        "kind": "port",
        "sample": f"{main_leg['workload']}, {main_leg['steps']} steps, OpenMP oracle ({main_leg['seconds']:.1f} s); one-core and cfg1 legs in `legs`",
        "cpu_model": cpu_model(),
        "threads_available": cores,
        "cores_source": cores_source,
        "cpus_in_affinity_mask": _CPUS_AT_START,
        "pinning": {"OMP_PROC_BIND": os.environ.get("OMP_PROC_BIND"), "OMP_PLACES": os.environ.get("OMP_PLACES")},
        "legs": legs,
    }


def cpu_baseline_in_child(nl: int):
    """Runs cpu_baseline() in a child process whose environment pins the OpenMP threads
    (OMP_PROC_BIND=close, OMP_PLACES=cores: BASELINE.md §3).  A child, because an OpenMP runtime
    reads these once when it starts and then binds the INITIAL thread too - in the bench process
    that would pin the thread that drives the GPU (and, under torch.distributed.run, the initial
    threads of all ranks to the same core).  The child never touches a GPU."""
    import subprocess

    env = dict(os.environ, OMP_PROC_BIND="close", OMP_PLACES="cores")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", str(nl)], env=env, capture_output=True, text=True, timeout=240)
    if r.returncode != 0:
        raise RuntimeError("CPU-baseline child failed: " + r.stderr[-2000:])
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def spawn_ranks(n: int, argv: list) -> int:
    """`python bench.py --gpus N` invoked plainly (N > 1): start the N ranks the way the driver
    does (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same
    arguments>`) as a child process, pass its stderr through, print rank 0's ONE JSON line on our
    stdout and return the child's exit code."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "1")             # the ranks are GPU-bound; torchrun would set it anyway (and say so)
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in child.stdout:
        t = ln.strip()
        if t.startswith("{") and t.endswith("}") and '"metric"' in t:
            line = t
        elif t:
            print(ln, end="", file=sys.stderr)  # anything else a rank wrote to stdout is not ours to publish
    rc = child.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        print("bench.py: the ranks exited without a result line", file=sys.stderr)
        rc = 1
    return rc


def pb_profile_from_product(pkg, p):
    """z profiles (phi, c, cn) of the Poisson-Boltzmann start, made by the PRODUCT:
    ekpnp_initialization_converged (the reference's Picard sweeps, LBM.cu:89-106, with a damping
    that converges on tall channels where PB_omega = 0.05 diverges) on a narrow 16x16 replica -
    the start is x-y uniform, so NX and NY do not enter.  Channels taller than 257 planes: the two
    double layers (lambda_D = 9.2 dz) are farther apart than exp(-128/9.2) ~ 1e-6 of the wall
    potential, so the replica has 257 planes and the planes in between carry its mid-plane values.
    Returns (profiles dict of [nz] arrays, note)."""
    nzr = p.nz if p.nz <= 257 else 257
    q = pkg.default_params(16, 16, nzr)
    for k in ("chargeinf", "voltage", "voltage2", "eps", "kB", "electron", "roomT", "convertCtoCharge", "PB_omega", "TH", "rho0", "dz"):
        setattr(q, k, getattr(p, k))
    q.Lz = (nzr - 1) * q.dz
    with pkg.Solver(q) as r:
        sweeps, res = r.initialization_converged(1e-9, 50000)
        prof = {k: r.get_field(k)[:, 0, 0].copy() for k in ("phi", "c", "cn")}
    if nzr != p.nz:
        h = nzr // 2
        prof = {k: np.concatenate([v[:h], np.full(p.nz - 2 * h, v[h]), v[nzr - h:]]) for k, v in prof.items()}
    note = (f"ekpnp_initialization_converged on a 16x16x{nzr} replica ({sweeps} PB sweeps, residual {res:.1e})"
            + (", mid-plane values between the two double layers" if nzr != p.nz else ""))
    return prof, note


def product_pb_state(sol, p, prof):
    """the fields gpu_initialization + the PB loop leave (LBM.cu:111-128,139-146), from z profiles"""
    nz, ny, nx = sol.shape
    z = (np.arange(nz) + sol.z0).astype(np.float64)
    sl = slice(sol.z0, sol.z0 + nz)
    col = lambda v: np.broadcast_to(np.asarray(v, dtype=np.float64)[:, None, None], sol.shape)  # noqa: E731
    sol.set_field("phi", col(prof["phi"][sl]))
    sol.set_field("c", col(prof["c"][sl]))
    sol.set_field("cn", col(prof["cn"][sl]))
    sol.set_field("rho", col(np.full(nz, p.rho0)))
    sol.set_field("T", col(p.TH * (p.Lz - p.dz * z) / p.Lz))  # LBM.cu:127
    zero = np.zeros(sol.shape)
    for k in ("ux", "uy", "uz", "Ex", "Ey", "Ez"):
        sol.set_field(k, zero)


def transport_or_exit(pkg, torch, dist, rank, want_native, allow_fallback):
    """The headline N>1 line runs on the library's own RCCL transport or not at all (fail closed).  EVERY rank asks the library
    whether it can bind RCCL (ekpnp_rccl_available: no device, no communicator), the ranks agree over the control plane, and if
    one cannot, all of them leave together with a non-zero exit and the library's message - before a context exists and before
    anybody is inside ncclCommInitRank.  Returns True when the native transport is to be used, False only under the explicit
    --allow-fallback-transport (the torch.distributed example transport runs, and the line is labelled)."""
    if not want_native:
        return False
    why = pkg.rccl_available()
    bad = bool(why)
    if dist is not None:
        flag = torch.tensor([1 if bad else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        bad = bool(int(flag.item()))
    if not bad:
        return True
    if allow_fallback and dist is not None:
        print(f"rank {rank}: the library's RCCL transport is unavailable ({why or 'on another rank'}); --allow-fallback-transport: "
              "the torch.distributed example transport runs instead", file=sys.stderr)
        return False
    if dist is not None:
        dist.destroy_process_group()
    raise SystemExit(f"bench.py: rank {rank}: the library's RCCL transport cannot be set up: {why or 'RCCL is unavailable on another rank'} "
                     "(--allow-fallback-transport would run the torch.distributed example transport instead, labelled as such)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="auto")
    ap.add_argument("--ic", default="perturbed", choices=["perturbed", "uniform"])
    ap.add_argument("--pb-iterations", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--in-place", action="store_true", help="one population buffer per lattice (ekpnp_params.in_place): 0.57x the memory of cfg3")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="N>1 data path.  nccl: the library's own RCCL transport (ekpnp_slab_attach_comm) - the only one a headline line "
                         "runs on.  gloo: the host-staged python transport of examples/host_transport.py, only to rehearse the multi-rank "
                         "flow on a one-GPU box; the line says so")
    ap.add_argument("--allow-fallback-transport", action="store_true",
                    help="opt-in safety net: if the library's RCCL communicator cannot be made, move the halos with the torch.distributed "
                         "example transport (examples/host_transport.py) and label the line FALLBACK TRANSPORT.  Without this flag such a "
                         "failure ends every rank with a non-zero exit and the library's message")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal on a one-GPU box: put every rank on device 0.  With --backend nccl each rank tells RCCL it is a "
                         "different host (NCCL_HOSTID), so the library's real communicator, ring and all-gather run between the "
                         "processes over RCCL's socket transport - functional coverage of the N>1 path, not a bandwidth figure")
    ap.add_argument("--force-slab", action="store_true",
                    help="N=1 only: run the multi-rank code path (split calls, comm stream, RCCL exchanges, the ring closing on the same rank)")
    ap.add_argument("--weak", action="store_true", help="the named grid is ONE RANK's slab; the channel is N of them (the 512^3-per-rank runs of rounds 1-2: --workload cfg3 --weak)")
    ap.add_argument("--scale-z", type=int, default=1, metavar="D", help="rehearsal: divide the z extent by D (several ranks sharing one GPU); the line says so")
    ap.add_argument("--no-batch-ab", action="store_true", help="skip the after-the-fact A/B of the opt-in knob batch_moments (config.batch_moments_ab)")
    ap.add_argument("--no-comm-ab", action="store_true", help="skip the after-the-fact knob A/B of the library's transport (N>1 and --force-slab lines: `comm_ab`)")
    ap.add_argument("--comm-ab-steps", type=int, default=10, help="steps per leg of that A/B")
    ap.add_argument("--after-deadline", type=float, default=300.0, metavar="SECONDS",
                    help="what runs after the timed region (knob A/Bs, copy probe, CPU baseline) gets this long; then the measured line is printed "
                         "as it stands, marked, and every rank leaves (class HeadlineGuard).  0: no deadline")
    ap.add_argument("--dry-run", action="store_true", help="check the launch plumbing only: rendezvous, barrier, one JSON line with the workload that WOULD run; no GPU")
    ap.add_argument("--cpu-baseline-only", type=int, default=None, metavar="LATTICES", help="internal: time the CPU oracle and print its JSON (the child of cpu_baseline_in_child)")
    args = ap.parse_args()
    if args.cpu_baseline_only is not None:
        print(json.dumps(cpu_baseline(args.cpu_baseline_only)), flush=True)
        return

    if args.single_device and args.gpus > 1:
        # several PROCESSES on one device oversubscribe its hardware queues unless each keeps to one (include/ekpnp.h:
        # ekpnp_plane_transforms; profiles/r05_shared_device_experiments.log).  Read by the HIP runtime when it starts: set here,
        # before torch is imported, in the launcher (the ranks inherit it) and in the ranks.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "1")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher.  Nothing has touched the GPU yet
        # (no torch import even); the ranks are CHILD processes, never an exec.
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...) or run it plainly")

    import torch

    # stdout is for the ONE JSON line.  C-level writers share fd 1 with it - gloo announces its connections there
    # ("[Gloo] Rank 0 is connected to 1 peer ranks ...", from EVERY rank, when the process group is made), RCCL prints a
    # version banner when its first communicator is made - so fd 1 points at stderr from here on, in every rank, and the
    # line goes out through the saved descriptor (HeadlineGuard / emit below).
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        data = (json.dumps(obj) + "\n").encode()
        while data:
            data = data[os.write(saved_stdout, data):]

    # control plane (rendezvous, barrier, max over ranks): torch.distributed over gloo, CPU tensors.
    # The DATA path of --backend nccl is RCCL inside libekpnp.so, not torch.
    dist = None
    if world > 1:
        import torch.distributed as dist  # noqa: WPS440

        dist.init_process_group("gloo")
    if args.dry_run:
        # every rank selects (and may reject) the workload BEFORE the first barrier: a bad grid ends all ranks with the same
        # message instead of leaving the others waiting for rank 0
        try:
            sel = select_workload(args.workload, world, 300 * 10**9, args.in_place, args.weak, args.scale_z)
        except ValueError as e:
            if dist is not None:
                dist.destroy_process_group()
            raise SystemExit(f"bench.py: {e}")
        # the transport decision of the real run, which needs no GPU either: all ranks leave non-zero if one cannot bind RCCL
        native_dry = transport_or_exit(G.load_package(), torch, dist, rank, (world > 1 or args.force_slab) and args.backend == "nccl",
                                       args.allow_fallback_transport)
        ipc_env = [os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")]
        if dist is not None:
            dist.barrier()
            ipc_env = [None] * world
            dist.all_gather_object(ipc_env, os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"))
        if rank == 0:
            emit({"metric": "MLUPS (full EK-PNP step)", "value": None, "unit": "MLUPS", "n_gpus": world, "dry_run": True, "scaling": sel["scaling"],
                              "env_by_rank": {"HSA_ENABLE_IPC_MODE_LEGACY": ipc_env},  # what the ranks of THIS launch mode run with
                              "transport": ("none (one context)" if world == 1 and not args.force_slab else
                                            "RCCL inside libekpnp.so" if native_dry else
                                            "torch.distributed example transport (rehearsal / FALLBACK)"),
                              "config": {"workload": sel["label"], "grid": list(sel["grid"]), "lattices": sel["lattices"], "nodes_per_rank": sel["nodes_per_rank"],
                                         "planes_per_rank": sel["planes_per_rank"], "scaling_note": sel["scaling_note"]}})
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    if args.single_device:
        local_rank = 0
    rehearsal = world > 1 and args.backend == "nccl" and args.single_device
    if rehearsal:
        # RCCL refuses two ranks of one host on one device; ranks that claim different hosts are
        # connected through its network (socket, loopback) transport instead.  Set before librccl loads.
        os.environ["NCCL_HOSTID"] = f"ekpnp-rehearsal-rank{rank}"
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_IB_DISABLE", "1")
    if world > 1 and int(os.environ.get("LOCAL_WORLD_SIZE", str(world))) == world:
        # every rank is on this node (the contract: N GPUs of ONE node): RCCL's bootstrap needs no network interface but the
        # loopback, whatever else the container shows it; the data path between the GPUs (P2P over xGMI) is not chosen by this.
        # Set before librccl loads; an explicit setting wins.  (Reported in the line: config.rccl_env.)
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    if args.single_device and world > 1:
        # ranks sharing one device must not race for its free memory: the placement search of ekpnp_create holds up to two
        # extra population arenas for a moment, and a rank that loses that race would leave its peers in ncclCommInitRank
        os.environ.setdefault("EKPNP_PLACEMENT_TRIES", "1")
        # (round 4 also forced rocFFT's plans here; round 5 found the cause - hardware-queue oversubscription, GPU_MAX_HW_QUEUES
        # above - and the own plane transforms run at full speed on a shared device too)
    ndev = torch.cuda.device_count()  # (counting devices does not initialise the GPU)
    nodev = local_rank >= ndev
    if dist is not None:  # the ranks leave together, before anybody waits for a rank that is gone
        flag = torch.tensor([1 if nodev else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        nodev = bool(int(flag.item()))
    if nodev:
        if dist is not None:
            dist.destroy_process_group()
        raise SystemExit(f"bench.py: rank {rank} (LOCAL_RANK {local_rank}) {'has no device of its own' if local_rank >= ndev else 'leaves with the others'}: {ndev} HIP device(s) visible for {world} ranks - "
                         "one rank per GPU, or --single-device for a functional rehearsal of the N>1 path on one GPU (labelled, not a scaling figure)")
    torch.cuda.set_device(local_rank)
    pkg = G.load_package()

    free_b, total_b = torch.cuda.mem_get_info()
    if args.single_device:
        free_b //= world  # the ranks share the one device
    if dist is not None:  # every rank must take the same in-place decision: the smallest free memory counts
        fb = torch.tensor([free_b], dtype=torch.int64)
        dist.all_reduce(fb, op=dist.ReduceOp.MIN)
        free_b = int(fb.item())
    try:
        sel = select_workload(args.workload, world, free_b, args.in_place, args.weak, args.scale_z)
    except ValueError as e:
        raise SystemExit(f"bench.py: {e}")
    wname, (nx, ny, nz_global), nl, use_in_place = sel["name"], sel["grid"], sel["lattices"], sel["in_place"]
    slab_path = world > 1 or args.force_slab
    fell_back = False  # True: --allow-fallback-transport and the library's own RCCL transport could not be set up: the example transport moved the halos
    native = transport_or_exit(pkg, torch, dist, rank, slab_path and args.backend == "nccl", args.allow_fallback_transport)
    if slab_path and args.backend == "nccl" and not native:
        fell_back = True
    p = pkg.default_params(nx, ny, nz_global)
    p.n_lattices = nl
    if nl < 4:
        p.Ra = 0.0
    if nl == 1:
        p.chargeinf, p.TH, p.exf = 0.0, 0.0, 1e9
    if args.pb_iterations is not None:
        p.pb_iterations = args.pb_iterations
    if use_in_place:
        p.in_place = 1

    def agree(err):
        """a rank that failed (e.g. out of memory) must not leave the others waiting in a collective"""
        if dist is None:
            if err is not None:
                raise SystemExit(f"bench.py: {err}")
            return
        flag = torch.tensor([0 if err is None else 1], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            dist.destroy_process_group()
            raise SystemExit(f"rank {rank}: set-up failed on at least one rank: {err}")

    # untimed start-up (the reference's timer also starts after it, main.cu:161-186)
    prof, ic_note = None, None
    if nz_global > 128 and nl > 1:
        prof, ic_note = pb_profile_from_product(pkg, p)  # before the big allocation, on this rank's own GPU

    err, runner, sol = None, None, None
    try:
        if not slab_path:
            sol = runner = pkg.Solver(p)
            transport = "none (one context)"
        elif native:
            sol = runner = pkg.Solver(p, rank, world, slab=True)
            transport = "RCCL inside libekpnp.so (ncclSend/ncclRecv ring + ncclAllGather, high-priority comm stream)"
            if rehearsal:
                transport += " - REHEARSAL: all ranks share device 0, RCCL socket transport; not a bandwidth figure"
        else:
            from examples.host_transport import DistributedSlab  # noqa: WPS433  (--backend gloo: the rehearsal transport, not product code)

            if dist is None:  # --force-slab --backend gloo
                import socket

                import torch.distributed as dist  # noqa: WPS440

                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    port = sk.getsockname()[1]
                dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
            if fell_back:  # --allow-fallback-transport and RCCL cannot be bound by the library: torch's own RCCL process group moves the halos
                runner = DistributedSlab(p, rank, world, dist, group=dist.new_group(backend="nccl"))
                transport = "torch.distributed RCCL process group (fallback: the library cannot bind RCCL)"
            else:
                runner = DistributedSlab(p, rank, world, dist)
                transport = "torch.distributed gloo, host-staged (rehearsal)"
            sol = runner.solver
    except Exception as e:  # noqa: BLE001
        err = e
    agree(err)
    if native:
        # ONE rank makes the RCCL id, the control plane hands it round, every rank attaches (collective)
        err = None
        try:
            ident = [pkg.comm_unique_id() if rank == 0 else None]
        except Exception as e:  # noqa: BLE001
            ident, err = [None], e
        if dist is not None:
            dist.broadcast_object_list(ident, src=0)
        if ident[0] is not None:
            try:
                sol.attach_comm(ident[0])
            except Exception as e:  # noqa: BLE001
                err = e
        failed = err is not None
        if dist is not None:
            flag = torch.tensor([1 if failed else 0], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            failed = bool(int(flag.item()))
        if failed and (dist is None or not args.allow_fallback_transport):
            # FAIL CLOSED: a headline line is only ever measured on the library's own transport.  Every rank learnt of the
            # failure through the all_reduce above and leaves here together, non-zero, with the library's message.
            if dist is not None:
                dist.destroy_process_group()
            raise SystemExit(f"bench.py: rank {rank}: the library's RCCL transport could not be set up"
                             f"{' on at least one rank' if err is None else ''}: {err if err is not None else 'see the other ranks'} "
                             "(--allow-fallback-transport would run the torch.distributed example transport instead, labelled as such)")
        if failed:
            # --allow-fallback-transport: the same slab contexts, halos moved by torch.distributed's own RCCL process
            # group (examples/host_transport.py) - the JSON line says which transport ran.  Every rank takes this branch together.
            print(f"rank {rank}: ekpnp_slab_attach_comm failed ({err}); --allow-fallback-transport: the torch.distributed RCCL example transport runs instead", file=sys.stderr)
            from examples.host_transport import DistributedSlab  # noqa: WPS433

            sol.close()
            data_group = dist.new_group(backend="nccl")
            runner = DistributedSlab(p, rank, world, dist, group=data_group)
            sol = runner.solver
            native = False
            fell_back = True
            transport = "torch.distributed RCCL process group (fallback: in-library communicator failed)"

    if prof is not None:
        product_pb_state(sol, p, prof)
    else:
        runner.initialization()
        ic_note = f"initialization() with {p.pb_iterations} PB sweeps"
    if args.ic == "perturbed":
        apply_perturbation(sol, None, p)
    runner.fast_Poisson()
    runner.init_equilibrium()

    def barrier():
        sol.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    runner.step(args.warmup)
    barrier()
    sol.kernel_timing(True)
    t0 = time.perf_counter()
    runner.step(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    n_launch, k_ms, k_nodes = sol.kernel_timing_get()
    n_stage_solves, stage_ms = sol.poisson_stage_timing_get() if slab_path else (0, None)  # (before phase_timing_get, which resets)
    n_solves, poisson_ms = sol.phase_timing_get()
    comm = None
    if native:
        raw = sol.comm_timing_get()
        waits = [sum(r["wait_ms"] for r in raw.values()) / max(1, args.steps)]
        if dist is not None:
            allw = [None] * world
            dist.all_gather_object(allw, waits[0])
            waits = allw
        comm = comm_block(raw, args.steps, 2 * 9 * nl * 8 * nx * ny, waits)
    sol.kernel_timing(False)

    # every rank's own phases (its own clock around the same barriers) and solve stages: rank 0 prints the spread
    my_phases = phases_of(dt, args.steps, k_ms, poisson_ms, n_solves)
    my_stages = None if not stage_ms else {k: round(v / max(1, n_stage_solves), 4) for k, v in stage_ms.items()}
    all_phases, all_stages = [my_phases], [my_stages]
    if dist is not None:
        all_phases, all_stages = [None] * world, [None] * world
        dist.all_gather_object(all_phases, my_phases)
        dist.all_gather_object(all_stages, my_stages)
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    rho = sol.get_field("rho")
    finite = bool(np.isfinite(rho).all())
    del rho
    nodes_total = nx * ny * nz_global
    mlups = nodes_total * args.steps / dt / 1e6

    # ---- the line as the timed region left it (rank 0), BEFORE anything else runs: HeadlineGuard keeps it safe -------------
    out = None
    if rank == 0:
        launches_per_step = max(1, n_launch // max(1, args.steps))
        k_avg_ms = k_ms / max(1, n_launch)
        bytes_per_launch = b_alg_lbm(nl) * k_nodes
        achieved = bytes_per_launch / (k_avg_ms * 1e-3) / 1e9 if k_avg_ms > 0 else 0.0
        # HBM traffic of one launch from the PMC counters: NOT measured by this run (counters need
        # rocprofv3 around the process) - the value of the committed profile of this workload, with
        # where it comes from; null if the profile does not cover this workload
        traffic, traffic_src, step_traffic = None, None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                rec = json.load(open(tpath)).get(wname, {})
                traffic = rec.get("hbm_bytes_per_launch")
                # what the whole step moves by the same counters (every kernel of a step x its launches): only for the
                # code path that profile ran - one context, two population buffers
                if not slab_path and not p.in_place:
                    step_traffic = step_traffic_of(rec)
                traffic_src = None if traffic is None else {"file": "profiles/pmc_traffic.json", "profiled_in": rec.get("round"),
                               "note": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE of that committed profile, NOT measured by this run; counted where requests "
                                       "leave the L2s - re-reads the 256 MiB Infinity Cache serves are included, so this bounds the HBM bytes from above"}
            except Exception:
                traffic = None
        out = {
            "metric": "MLUPS (full EK-PNP step)",
            "value": round(mlups, 2),
            "unit": "MLUPS",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": sel["scaling"],
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": sel["label"] + (", multi-rank code path on one rank (ring to itself)" if world == 1 and slab_path else "")
                + (" - FALLBACK TRANSPORT (torch.distributed, not the library's)" if fell_back else ""),
                "grid": [nx, ny, nz_global],
                "nodes_per_rank": sel["nodes_per_rank"],
                "planes_per_rank": sel["planes_per_rank"],
                "scaling_note": sel["scaling_note"],
                "lattices": nl,
                "in_place": bool(p.in_place),
                "transport": transport,
                "ic": ic_note + (" + closed-form 3-D perturbation" if args.ic == "perturbed" else ""),
                "b_alg_step_bytes_per_node": b_alg_step(nl),
                "step_roofline_frac": round(b_alg_step(nl) * mlups / world * 1e6 / (HBM_PEAK_GBS * 1e9), 4),
                # the yardstick above credits SURVEY 8(d)'s compulsory bytes (1 856 B/node at cfg3).  What the step REALLY
                # moves, summed over all its kernels from the committed counter profile of this workload (NOT measured by
                # this run; null without such a profile), and the HBM rate that is at this run's step time
                "step_traffic_bytes": None if step_traffic is None else int(step_traffic),
                "step_hbm_GBps": None if step_traffic is None else round(step_traffic / (dt / args.steps) / 1e9, 1),
                "step_traffic_over_algorithmic": None if step_traffic is None else round(step_traffic / (b_alg_step(nl) * nodes_total), 4),
                "device_bytes": sol.device_bytes(),
                # arenas the context timed at creation and the one it kept (tried 0: the arena is most of the device - cfg3, cfg5)
                "placement": sol.placement_report(),
                # the library's own row / column passes or rocFFT plans; ranks of this lattice on rank 0's device (> 1: a rehearsal)
                "plane_transforms": sol.plane_transforms(),
                # the cache-aware orders in effect (include/ekpnp.h: ekpnp_pass_order): rows per band of the interior sweep, kx blocks of the solve
                "pass_order": sol.pass_order(),
                # what RCCL was told through the environment (N>1; rehearsals add NCCL_HOSTID: ranks on one device pose as hosts)
                "rccl_env": {k: v for k, v in sorted(os.environ.items()) if k.startswith(("NCCL_", "RCCL_")) or k in ("HSA_ENABLE_IPC_MODE_LEGACY", "GPU_MAX_HW_QUEUES")} if slab_path else None,
                "finite": finite,
                # HIP events on the context's stream inside the timed region: the collide sweep of the interior
                # planes, the Poisson solve (on slabs: stage 1 to stage 3, exchanges included), and what is
                # left of the step (wall planes, halo pack / unpack, dependency gaps)
                "phases_ms_per_step": phases_of(dt, args.steps, k_ms, poisson_ms, n_solves),
                "batch_moments_ab": None,
            },
            "roofline": {
                "kernel": "k_collide_bulk",
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_source": traffic_src,
                "copy_GBps": None,
                "frac_of_copy": None,
                "bytes_per_node": b_alg_lbm(nl),
                "nodes_per_launch": int(k_nodes),
                "launches_per_step": launches_per_step,
                "avg_launch_ms": round(k_avg_ms, 4),
            },
        }
        if slab_path:
            # a line measured on the python safety-net transport must not pass for the library's: said at the top level
            out["transport_fallback"] = fell_back
            # the scaling loss, itemised: what the compute stream waited for, what the exchanges took, what they moved
            out["comm"] = comm if comm is not None else {"source": "not measured: the python example transport (examples/host_transport.py) ran, not the library's"}
            # the spread over the ranks (each rank's own events and clock; rank 0's own values are config.phases_ms_per_step)
            out["config"]["phases_ms_per_step_by_rank"] = min_max_by_rank(all_phases)
            if all(st is not None for st in all_stages):
                # where a solve's time goes on a slab: stage 1 | EDGE all-gather as the compute stream saw it | stage 2 | PHI exchange | stage 3
                out["config"]["poisson_stages_ms_per_solve_by_rank"] = min_max_by_rank(all_stages)

    # Everything below runs after the timed region and outside `value`.  It must not be able to lose the line above: every
    # rank arms the guard now (class HeadlineGuard), each leg is tried on its own, and the line goes out whatever they do.
    sys.stdout.flush()
    guard = HeadlineGuard(rank, args.after_deadline, fd=saved_stdout)
    guard.arm(out)
    after_errors = {}

    def leg(name, fn):
        """one after-the-fact leg: named to the guard, never fatal to the line (an exception is recorded under its name)"""
        guard.update(leg=name)
        try:
            return fn()
        except Exception as e:  # noqa: BLE001
            after_errors[name] = f"{type(e).__name__}: {e}"
            print(f"bench.py: rank {rank}: after-the-fact leg '{name}' failed: {after_errors[name]}", file=sys.stderr)
            return None

    if rank == 0:
        # secondary denominator: what a plain contiguous copy reaches on THIS device, now
        def copy_probe():
            free_now, _ = torch.cuda.mem_get_info()
            return sol.copy_bandwidth(min(2 << 30, free_now // 4))  # (may not fit next to a 276 GB lattice: then the leg reports that)

        copy_gbs = leg("copy_bandwidth", copy_probe)
        if copy_gbs:
            out["roofline"]["copy_GBps"] = round(copy_gbs, 1)
            out["roofline"]["frac_of_copy"] = round(out["roofline"]["achieved"] / copy_gbs, 4)
        guard.update(out)

    # The opt-in knob "batch_moments" (include/ekpnp.h) - inside one step(n) call only the last step stores rho, u, c, cn, T.
    # The HEADLINE never uses it (every step of the timed region stores them, as the reference's step does, LBM.cu:807-813);
    # this A/B says what a host that steps in batches between its outputs gets.
    if not args.no_batch_ab and (native or not slab_path):
        n_ab = max(10, min(args.steps, 40))

        def timed_ms(n):
            barrier()
            t_ = time.perf_counter()
            runner.step(n)
            barrier()
            v = (time.perf_counter() - t_) / n * 1e3
            if dist is not None:
                tv = torch.tensor([v], dtype=torch.float64)
                dist.all_reduce(tv, op=dist.ReduceOp.MAX)
                v = float(tv.item())
            return v

        def batch_leg():
            every = timed_ms(n_ab)
            sol.tune("batch_moments", 1)
            try:
                last_only = timed_ms(n_ab)
            finally:
                sol.tune("batch_moments", 0)
            return {"steps_per_call": n_ab, "every_step_stores_ms_per_step": round(every, 4), "last_step_stores_ms_per_step": round(last_only, 4),
                    "every_step_stores_MLUPS": round(nx * ny * nz_global / every / 1e3, 1), "last_step_stores_MLUPS": round(nx * ny * nz_global / last_only / 1e3, 1),
                    "note": "after the timed region, NOT the headline: ekpnp_tune(ctx, \"batch_moments\", 1) - inside one step(n) call only the last "
                            "step stores the seven moment arrays (56 of the sweep's 1 808 B/node); same visible bits; HIP timing hooks off in both legs"}

        batch_ab = leg("batch_moments_ab", batch_leg)
        if out is not None:
            out["config"]["batch_moments_ab"] = batch_ab
            guard.update(out)

    # A few steps under each knob that one GPU cannot decide, on the live contexts (ekpnp_tune), so that ONE multi-GPU run
    # says which default is right on xGMI.  Every rank runs every leg.
    if native and not args.no_comm_ab and args.comm_ab_steps > 0:
        base = ab_baseline()
        comm_ab = []
        for label, knobs in COMM_AB_LEGS:
            r = leg(f"comm_ab: {label}", lambda: comm_ab_leg(label, knobs, sol, runner, args.comm_ab_steps, barrier, dist, torch, world, base))  # noqa: B023
            comm_ab.append(r if r is not None else {"knob": label, "error": after_errors.get(f"comm_ab: {label}", "failed")})
            if out is not None:
                out["comm_ab"] = {"note": f"after the timed region, outside `value`: {args.comm_ab_steps} steps per leg on the live contexts "
                                          "(ekpnp_tune on every rank); ms_per_step and the waits are maxima over the ranks; the timed region ran with `baseline`",
                                  "baseline": ab_baseline(), "legs": list(comm_ab)}
                guard.update(out)

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            def cpu_leg():
                runner.close()
                return cpu_baseline_in_child(nl)

            cb = leg("cpu_baseline", cpu_leg)
            out["cpu_baseline"] = cb if cb is not None else {"error": after_errors.get("cpu_baseline"), "kind": "port", "value": None, "unit": "MLUPS", "cores": None,
                                                             "sample": "not measured: the CPU-baseline child failed (message in `error`)"}
        if after_errors:
            out["after_the_fact"] = {"status": "legs failed", "errors": after_errors}
        sys.stdout.flush()
    guard.finish(out)  # rank 0: the ONE JSON line, on the real stdout; every rank: disarm
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
