"""GPU tests of the z-slab path: P slabs must reproduce the single-context
result - the LBM part bit for bit (same kernels, same arithmetic per node), the Poisson part to
rounding (the distributed tridiagonal associates differently)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _single(pkg, O, p, start, steps):
    with pkg.Solver(p) as s:
        s.initialization()
        init = s.fields()
        st = start(init)
        s.set_fields(st)
        s.fast_Poisson()
        s.init_equilibrium()
        s.step(steps)
        return init, st, s.fields()


@pytest.mark.parametrize("shape,nslabs", [((16, 12, 16), 2), ((20, 6, 24), 3), ((70, 5, 32), 4), ((16, 8, 64), 8), ((12, 6, 40), 2), ((10, 4, 16), 4), ((8, 4, 102), 3),
                                          ((18, 6, 21), 1)])
def test_local_slab_group_equals_single_context(pkg, O, shape, nslabs):
    from examples.host_transport import LocalSlabGroup

    p = pkg.default_params(*shape)
    p.pb_iterations = 15
    po = O.default_params(*shape)
    init1, st, want = _single(pkg, O, p, lambda f: O.perturb_fields(po, f), 7)
    g = LocalSlabGroup(p, nslabs)
    try:
        g.initialization()
        init = g.fields()
        e0 = O.rel_l2(init, init1, {k: v for k, v in O.GROUPS.items() if k != "u"})
        assert max(e0.values()) < 1e-12, e0
        g.set_fields(st)
        g.fast_Poisson()
        g.init_equilibrium()
        g.step(7)
        got = g.fields()
    finally:
        g.close()
    err = O.rel_l2(got, want)
    assert all(v < (1e-7 if k == "u" else 1e-11) for k, v in err.items()), err


@pytest.mark.parametrize("shape,nslabs", [((16, 12, 16), 2), ((20, 6, 72), 3), ((12, 4, 160), 2)])
def test_in_place_slabs_equal_single_context(pkg, O, shape, nslabs):
    """in_place = 1 on slab contexts: the first/last plane of a slab are collided into a staging
    buffer (they feed the halo exchange before the ordered sweep of the planes in between)."""
    from examples.host_transport import LocalSlabGroup

    p = pkg.default_params(*shape)
    p.pb_iterations = 10
    po = O.default_params(*shape)
    _, st, want = _single(pkg, O, p, lambda f: O.perturb_fields(po, f), 7)
    p.in_place = 1
    g = LocalSlabGroup(p, nslabs)
    try:
        g.initialization()
        g.set_fields(st)
        g.fast_Poisson()
        g.init_equilibrium()
        g.step(7)
        got = g.fields()
        mem = sum(s.device_bytes() for s in g.sol)
    finally:
        g.close()
    err = O.rel_l2(got, want)
    assert all(v < (1e-7 if k == "u" else 1e-11) for k, v in err.items()), err
    for k in ("rho", "c", "cn", "T", "ux", "uy", "uz"):  # the LBM part is the same arithmetic per node
        assert np.abs(got[k] - want[k]).max() <= 1e-9 * np.abs(want[k]).max(), k


def test_slabs_three_lattices(pkg, O):
    from examples.host_transport import LocalSlabGroup

    shape = (24, 6, 20)
    p = pkg.default_params(*shape)
    p.pb_iterations, p.Ra, p.n_lattices = 10, 0.0, 3
    po = O.default_params(*shape)
    _, st, want = _single(pkg, O, p, lambda f: O.perturb_fields(po, f), 5)
    g = LocalSlabGroup(p, 2)
    try:
        g.initialization()
        g.set_fields(st)
        g.fast_Poisson()
        g.init_equilibrium()
        g.step(5)
        got = g.fields()
    finally:
        g.close()
    err = O.rel_l2(got, want, {k: v for k, v in O.GROUPS.items() if k != "T"})
    assert all(v < (1e-7 if k == "u" else 1e-11) for k, v in err.items()), err


def _oracle_from(O, po, st, steps):
    """the ORACLE (CPU restatement of the reference's step) from the same start fields: fast_Poisson, init_equilibrium, steps"""
    orc = O.Oracle(po)
    try:
        orc.set_fields(st)
        orc.fast_poisson()
        orc.init_equilibrium()
        orc.step(steps)
        return orc.fields(), orc.current(), orc.umax()
    finally:
        orc.close()


def _assert_vs_oracle(O, got, oracle_result, ranks):
    want, cur, um = oracle_result
    err = O.rel_l2(got, want)
    assert all(v <= (1e-7 if k == "u" else 1e-9) for k, v in err.items()), ("vs oracle", err)
    for d in ranks:
        assert abs(float(d["current"]) - cur) <= 1e-8 * abs(cur)
        assert abs(float(d["umax"]) - um) <= 1e-6 * abs(um) + 1e-30


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("shape,nprocs,in_place,backend", [((16, 12, 16), 2, 0, "gloo"), ((48, 20, 48), 4, 0, "gloo"),
                                                          ((24, 8, 48), 2, 1, "gloo"), ((32, 16, 40), 1, 0, "nccl"),
                                                          ((32, 16, 40), 1, 1, "nccl"), ((24, 8, 36), 3, 0, "nccl")])
def test_processes_sharing_one_gpu(pkg, O, tmp_path, shape, nprocs, in_place, backend):
    """The real multi-process path (DistributedSlab + RingTransport), launched like the driver
    launches bench.py: 2 and 4 ranks sharing the one GPU of the box over gloo (host-staged), and
    ONE rank over RCCL ("nccl"): a single slab whose ring neighbours are itself, so every halo,
    all-gather and phi exchange of the multi-GPU step really goes through RCCL send/recv on
    tensors aliasing the library's device buffers; and THREE ranks over RCCL, each claiming its own
    host id so that RCCL accepts them on one device (socket transport; see _slab_worker.py)."""
    p = pkg.default_params(*shape)
    p.pb_iterations = 12
    po = O.default_params(*shape)
    _, st, want = _single(pkg, O, p, lambda f: O.perturb_fields(po, f), 6)
    with pkg.Solver(p) as ref:  # diagnostics of the single-context result
        ref.set_fields(want)
        want_current, want_umax = ref.current(), ref.umax()
    np.savez(tmp_path / "start.npz", **st)
    env = dict(os.environ, EKPNP_SLAB_OUT=str(tmp_path), EKPNP_SLAB_IN_PLACE=str(in_place), EKPNP_SLAB_BACKEND=backend, EKPNP_SLAB_GRID="x".join(map(str, shape)), OMP_NUM_THREADS="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nprocs}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_slab_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    parts = sorted((np.load(tmp_path / f"rank{k}.npz") for k in range(nprocs)), key=lambda d: int(d["z0"]))
    got = {k: np.concatenate([d[k] for d in parts], axis=0) for k in O.FIELDS}
    err = O.rel_l2(got, want)
    assert all(v < (1e-7 if k == "u" else 1e-11) for k, v in err.items()), err
    for d in parts:  # every rank holds the combined diagnostics
        assert abs(float(d["current"]) - want_current) <= 1e-9 * abs(want_current)
        assert abs(float(d["umax"]) - want_umax) <= 1e-6 * abs(want_umax) + 1e-30


@pytest.mark.parametrize("shape,nprocs,in_place", [((16, 12, 16), 2, 0), ((24, 10, 23), 3, 0), ((32, 8, 48), 4, 1)])
def test_native_rccl_ranks_sharing_one_gpu(pkg, O, tmp_path, shape, nprocs, in_place):
    """The library's OWN RCCL transport with MORE THAN ONE rank: one process per rank, each with a
    communicator made by ncclCommInitRank from the id rank 0 created (ekpnp_comm_unique_id ->
    ekpnp_slab_attach_comm), halo ring by ncclSend/ncclRecv, interface coefficients by ncclAllGather,
    diagnostics by ncclAllReduce, whole-lattice files by the ranks taking turns.  RCCL refuses two ranks
    of one HOST on one device, so every rank claims its own host id (NCCL_HOSTID) and RCCL wires them
    through its socket transport - the library's side is the multi-GPU code path unchanged.  Checked:
    fields vs the single-context run (uneven slabs included), combined diagnostics on every rank, the
    data_end / Tecplot files byte for byte against the single context's, bitwise continuation from
    per-rank checkpoints, and the reference's restart route through the collective reader."""
    p = pkg.default_params(*shape)
    p.pb_iterations = 12
    po = O.default_params(*shape)
    _, st, want = _single(pkg, O, p, lambda f: O.perturb_fields(po, f), 6)
    with pkg.Solver(p) as ref:
        ref.set_fields(want)
        want_current, want_umax = ref.current(), ref.umax()
        ref.save_data_end(str(tmp_path / "single_end.dat"), 0.25)
        ref.save_data_tecplot(str(tmp_path / "single_tec.dat"), 0.25)
    np.savez(tmp_path / "start.npz", **st)
    env = dict(os.environ, EKPNP_SLAB_OUT=str(tmp_path), EKPNP_SLAB_IN_PLACE=str(in_place), EKPNP_SLAB_GRID="x".join(map(str, shape)),
               OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if nprocs == 3:  # the three-rank case takes the A/B partner of k_collide_faces: a launch per face (k_collide_wall, k_collide_edge)
        env["EKPNP_MERGED_FACES"] = "0"
        env["EKPNP_EDGE_P2P"] = "1"  # ... and gathers the edge values with direct send / receive pairs (two peers per rank) from the first solve on
    if nprocs == 4:  # round 5: the four-rank case gathers the edge values in 3 pipelined mode blocks from the first solve on (nxh = 24)
        env["EKPNP_EDGE_CHUNKS"] = "3"
    if nprocs == 2:  # ... and the two-rank case turns the transport's knobs on the live communicator after half of the steps
        env["EKPNP_RCCL_TUNE_MIDWAY"] = "inline_exchanges=0,edge_chunks=2,lead_planes=0,comm_cus=8,edge_p2p=1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nprocs}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_rccl_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-4000:]
    parts = sorted((np.load(tmp_path / f"rank{k}.npz") for k in range(nprocs)), key=lambda d: int(d["z0"]))
    got = {k: np.concatenate([d[k] for d in parts], axis=0) for k in O.FIELDS}
    err = O.rel_l2(got, want)
    assert all(v < (1e-7 if k == "u" else 1e-11) for k, v in err.items()), err
    _assert_vs_oracle(O, got, _oracle_from(O, po, st, 6), parts)  # the ranks' result against the ORACLE, not only the HIP single context
    for d in parts:
        assert abs(float(d["current"]) - want_current) <= 1e-9 * abs(want_current)
        assert abs(float(d["umax"]) - want_umax) <= 1e-6 * abs(want_umax) + 1e-30
        assert bool(d["ckpt_same"]) and float(d["t_ck"]) == pytest.approx(6 * p.dt, rel=1e-12)
        assert float(d["t_read"]) == pytest.approx(0.25, rel=1e-5)  # the file holds the time as %10.6f-style text
        assert str(d["io_error"]) != "" and np.isfinite(float(d["umax_after"]))  # a failed open reaches every rank; no hang
    # the files the ranks wrote in turns: text of the SAME numbers as the single context's whenever the fields agree to
    # the printed digits; the slab fields equal the single context's to ~1e-13, so compare parsed values, then sizes
    for mine, single in (("data_end.dat", "single_end.dat"), ("tec.dat", "single_tec.dat")):
        a, b = (tmp_path / mine).read_text().splitlines(), (tmp_path / single).read_text().splitlines()
        assert len(a) == len(b) and len(a) >= np.prod(shape), (mine, len(a), len(b))
        skip = len(a) - int(np.prod(shape))
        assert a[:skip] == b[:skip], (mine, a[:skip], b[:skip])
        va = np.array([x.split() for x in a[skip:]], dtype=np.float64)
        vb = np.array([x.split() for x in b[skip:]], dtype=np.float64)
        scale = np.abs(vb).max(axis=0) + 1e-300
        assert (np.abs(va - vb) / scale).max() < 1e-5, (mine, (np.abs(va - vb) / scale).max())
    # restart through the collective reader: the fields the ranks hold afterwards are the file's (text precision);
    # interior planes only - the writer extrapolates rho, c, cn, u onto the plates (LBM.cu:2527-2542)
    re = {k: np.concatenate([d["re_" + k] for d in parts], axis=0) for k in ("rho", "c", "cn", "T")}
    for k, v in re.items():
        assert np.abs(v[1:-1] - want[k][1:-1]).max() <= 2e-6 * max(1.0, np.abs(want[k]).max()), k


@pytest.mark.parametrize("own_fft", [None, "0"])
def test_native_rccl_two_ranks_full_width_planes(pkg, O, tmp_path, own_fft):
    """Two real RCCL ranks (as above) on planes of cfg3's full width: 512 x 512 x 24, i.e. 8 tiles per row,
    the two-node phi / E kernel, 37.7 MB halo messages per direction and lattice group - against the single
    context on the same lattice (fields and combined diagnostics).
    Round 5: the two ranks SHARE the box's device, which the library finds out when the communicator is made
    (ekpnp_plane_transforms: ranks_on_device == 2); the workers keep to one hardware queue each (GPU_MAX_HW_QUEUES=1,
    include/ekpnp.h).  Both plane transforms: the library's own row / column passes (the default on 512-wide planes; the
    midway tune cuts their column pass into 4 mode blocks) and rocFFT's plans (EKPNP_OWN_FFT=0)."""
    shape, nprocs = (512, 512, 24), 2
    p = pkg.default_params(*shape)
    p.pb_iterations = 12
    po = O.default_params(*shape)
    _, st, want = _single(pkg, O, p, lambda f: O.perturb_fields(po, f), 6)
    with pkg.Solver(p) as ref:
        ref.set_fields(want)
        want_current, want_umax = ref.current(), ref.umax()
    np.savez(tmp_path / "start.npz", **st)
    env = dict(os.environ, EKPNP_SLAB_OUT=str(tmp_path), EKPNP_SLAB_IN_PLACE="0", EKPNP_SLAB_GRID="x".join(map(str, shape)),
               EKPNP_RCCL_FIELDS_ONLY="1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               EKPNP_RCCL_TUNE_MIDWAY="edge_chunks=4")  # round 5: the last three steps with the column pass and the all-gather in 4 mode blocks
    env.pop("EKPNP_OWN_FFT", None)
    if own_fft is not None:
        env["EKPNP_OWN_FFT"] = own_fft
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nprocs}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_rccl_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    parts = sorted((np.load(tmp_path / f"rank{k}.npz") for k in range(nprocs)), key=lambda d: int(d["z0"]))
    got = {k: np.concatenate([d[k] for d in parts], axis=0) for k in O.FIELDS}
    err = O.rel_l2(got, want)
    assert all(v < (1e-7 if k == "u" else 1e-11) for k, v in err.items()), err
    for d in parts:
        assert abs(float(d["current"]) - want_current) <= 1e-9 * abs(want_current)
        assert abs(float(d["umax"]) - want_umax) <= 1e-6 * abs(want_umax) + 1e-30
        assert int(d["ranks_on_device"]) == 2 and bool(d["own_passes"]) == (own_fft is None)
    if own_fft is None:
        _assert_vs_oracle(O, got, _oracle_from(O, po, st, 6), parts)  # 6.3 M nodes x 6 steps on the host cores


def test_bench_under_the_drivers_launcher_two_rccl_ranks(tmp_path):
    """bench.py ITSELF, started the way the driver starts an N>1 run (`python -m torch.distributed.run --nnodes=1
    --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 --steps K --warmup W`), with two real RCCL
    ranks of the library's transport on the box's one device (--single-device: the functional rehearsal, not a bandwidth
    figure).  The first run on a multi-GPU node cannot be repeated cheaply, so the whole line is checked here: exit 0, ONE
    JSON line on stdout and nothing else, the driver's keys, `comm`, per-rank phases and solve stages, every `comm_ab` leg
    without an error, `batch_moments_ab`, no `after_the_fact` mark (class HeadlineGuard stayed quiet), finite fields."""
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)  # bench.py must set it for the ranks of this launch mode by itself
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--single-device", "--workload", "128x128x48", "--comm-ab-steps", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    out = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(out) == 1, r.stdout[-2000:]  # RCCL's banner and everything else went to stderr
    d = json.loads(out[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 2 and d["dtype"] == "f64" and d["value"] > 0
    assert d["transport_fallback"] is False and d["config"]["transport"].startswith("RCCL inside libekpnp.so")
    assert d["config"]["planes_per_rank"] == [24, 24] and d["config"]["finite"] is True
    assert "after_the_fact" not in d
    assert d["comm"]["halo"]["exchanges_per_step"] == 1 and len(d["comm"]["wait_ms_per_step_by_rank"]) == 2
    assert set(d["config"]["phases_ms_per_step_by_rank"]) == {"min", "max"}
    assert set(d["config"]["poisson_stages_ms_per_solve_by_rank"]["max"]) == {"stage1", "edge_exchange", "stage2", "phi_exchange", "stage3"}
    legs = d["comm_ab"]["legs"]
    assert [g["knob"] for g in legs][0] == "defaults" and len(legs) >= 5
    assert not [g for g in legs if "error" in g], legs
    assert d["config"]["batch_moments_ab"]["last_step_stores_ms_per_step"] > 0
    assert d["config"]["plane_transforms"]["ranks_on_device"] == 2
