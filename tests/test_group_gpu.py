"""GPU tests of the library's own z-slab transport (csrc/slab_team.hip) through the C ABI:
ekpnp_group_* (one process, P slabs) and ekpnp_slab_attach_comm (one process per rank).

The comparisons are against the ORACLE (the CPU restatement of the reference's step), not against
the single-context HIP run: grids 16x12x16, 70x5x32, 130x4x64, P = 2, 4, 8, two-buffer and
in-place populations.  A one-GPU box cannot hold two RCCL ranks on its device (RCCL refuses), so
multi-slab groups there move their halos by device copies (EKPNP_TRANSPORT_COPY: same events,
same comm streams, same call order) and RCCL is exercised with ONE rank whose ring closes on
itself: every halo, all-gather and phi exchange then goes through ncclSend/ncclRecv/ncclAllGather
inside the library."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL, TOL_U = 1e-9, 1e-7


def _mirror(pkg, po):
    p = pkg.Params()
    for name, _ in p._fields_:
        setattr(p, name, getattr(po, name))
    return p


def _oracle_run(O, po, steps):
    orc = O.Oracle(po)
    orc.initialization()
    init = orc.fields()
    start = O.perturb_fields(po, init)
    orc.set_fields(start)
    orc.fast_poisson()
    pois = orc.fields()
    orc.init_equilibrium()
    orc.step(steps)
    out = orc.fields()
    cur, um = orc.current(), orc.umax()
    orc.close()
    return init, start, pois, out, cur, um


def _check(O, got, want, groups=None, where=""):
    err = O.rel_l2(got, want, groups or O.GROUPS)
    bad = {k: v for k, v in err.items() if not v <= (TOL_U if k == "u" else TOL)}
    assert not bad, (where, err)


def _drive(O, run, po, ref, steps):
    """the same call sequence on anything that speaks the reference's verbs"""
    init, start, pois, out, cur, um = ref
    run.initialization()
    _check(O, run.fields(), init, {k: v for k, v in O.GROUPS.items() if k != "u"}, "init")
    run.set_fields(start)
    run.fast_Poisson()
    _check(O, run.fields(), pois, {"phi": ["phi"], "E": ["Ex", "Ey", "Ez"]}, "poisson")
    run.init_equilibrium()
    run.step(steps)
    _check(O, run.fields(), out, where=f"step {steps}")
    assert abs(run.current() - cur) <= 1e-8 * abs(cur)
    assert abs(run.umax() - um) <= 1e-6 * abs(um) + 1e-30


CASES = [((16, 12, 16), 2), ((16, 12, 16), 4), ((70, 5, 32), 2), ((70, 5, 32), 4), ((70, 5, 32), 8), ((130, 4, 64), 2), ((130, 4, 64), 4),
         ((130, 4, 64), 8)]


@pytest.mark.parametrize("in_place", [0, 1])
@pytest.mark.parametrize("shape,nslabs", CASES)
def test_group_of_slabs_vs_oracle(pkg, O, shape, nslabs, in_place):
    po = O.default_params(*shape)
    po.pb_iterations = 15
    ref = _oracle_run(O, po, 7)
    p = _mirror(pkg, po)
    p.in_place = in_place
    with pkg.Group(p, nslabs, devices=[0] * nslabs) as g:
        assert g.transport == pkg.TRANSPORT_COPY and g.n == nslabs
        assert [g.slab_extent(i) for i in range(nslabs)] == [(i * shape[2] // nslabs, shape[2] // nslabs) for i in range(nslabs)]
        _drive(O, g, po, ref, 7)


@pytest.mark.parametrize("in_place", [0, 1])
def test_one_rank_rccl_group_vs_oracle(pkg, O, in_place):
    """EKPNP_TRANSPORT_RCCL with one slab: ncclCommInitAll on one device, the ring closes on the rank."""
    po = O.default_params(70, 5, 32)
    po.pb_iterations = 15
    ref = _oracle_run(O, po, 7)
    p = _mirror(pkg, po)
    p.in_place = in_place
    with pkg.Group(p, 1, devices=[0], transport=pkg.TRANSPORT_RCCL) as g:
        assert g.transport == pkg.TRANSPORT_RCCL
        _drive(O, g, po, ref, 7)


def test_attached_comm_makes_the_reference_verbs_work_on_a_slab(pkg, O):
    """One process per rank as bench.py runs it: ekpnp_comm_unique_id on one rank, every rank
    ekpnp_slab_attach_comm (ncclCommInitRank), then the plain verbs on the slab context.  One rank
    here (a second rank would need a second GPU)."""
    po = O.default_params(48, 6, 24)
    po.pb_iterations = 12
    ref = _oracle_run(O, po, 6)
    p = _mirror(pkg, po)
    s = pkg.Solver(p, 0, 1, slab=True)
    try:
        with pytest.raises(pkg.EkpnpError, match="transport"):
            s.step(1)  # no transport yet: refuses instead of computing something else
        s.attach_comm(pkg.comm_unique_id())
        with pytest.raises(pkg.EkpnpError, match="already"):
            s.attach_comm(pkg.comm_unique_id())
        _drive(O, s, po, ref, 6)
        n, res = s.initialization_converged(1e-9, 2000)
        assert 0 < n < 2000 and res <= 1e-9
    finally:
        s.close()


def test_group_rejects_what_it_cannot_do(pkg):
    p = pkg.default_params(16, 8, 16)
    with pytest.raises(pkg.EkpnpError, match="RCCL"):
        pkg.Group(p, 2, devices=[0, 0], transport=pkg.TRANSPORT_RCCL)
    with pytest.raises(pkg.EkpnpError):
        pkg.Group(p, 5, devices=[0] * 5)  # 16 planes / 5 slabs
    with pytest.raises(pkg.EkpnpError):
        pkg.Group(p, 2, devices=[0, 99])
    with pkg.Group(p, 2, devices=[0, 0]) as g:
        import ctypes as C

        h = g.slab_handle(0)
        assert pkg.load_library().ekpnp_step(h, 1) != 0  # a member is driven through the group only
        assert b"ekpnp_group" in pkg.load_library().ekpnp_last_error(h)


def test_group_files_and_diagnostics_are_the_single_context_ones(pkg, O, tmp_path):
    """data.dat / data_end.dat / umax.dat written by a 4-slab group are byte-identical to a single
    context's (LBM.cu:2492-2630,2748), read_data and the EKPNPST1 state file are interchangeable
    between the two, current() and umax agree."""
    shape = (20, 6, 24)
    p = pkg.default_params(*shape)
    p.pb_iterations = 10
    po = O.default_params(*shape)
    with pkg.Solver(p) as s:
        s.initialization()
        s.set_fields(O.perturb_fields(po, s.fields()))
        s.fast_Poisson(); s.init_equilibrium(); s.step(5)
        f1 = s.fields()
        s.save_data_tecplot(str(tmp_path / "a.dat"), 1.5e-9, first=True)
        s.save_data_tecplot(str(tmp_path / "a.dat"), 2.5e-9, first=False, append=True)
        s.save_data_end(str(tmp_path / "a_end.dat"), 2.5e-9)
        s.record_umax(str(tmp_path / "a_umax.dat"), 2.5e-9, append=False)
        s.save_state(str(tmp_path / "a.bin"), 2.5e-9)
        cur1, um1 = s.current(), s.umax()
    with pkg.Group(p, 4, devices=[0] * 4) as g:
        g.set_fields(f1)
        g.save_data_tecplot(str(tmp_path / "b.dat"), 1.5e-9, first=True)
        g.save_data_tecplot(str(tmp_path / "b.dat"), 2.5e-9, first=False, append=True)
        g.save_data_end(str(tmp_path / "b_end.dat"), 2.5e-9)
        g.record_umax(str(tmp_path / "b_umax.dat"), 2.5e-9, append=False)
        g.save_state(str(tmp_path / "b.bin"), 2.5e-9)
        assert g.current() == cur1 and g.umax() == um1
        for a, b in (("a.dat", "b.dat"), ("a_end.dat", "b_end.dat"), ("a_umax.dat", "b_umax.dat"), ("a.bin", "b.bin")):
            assert (tmp_path / a).read_bytes() == (tmp_path / b).read_bytes(), (a, b)
        # restart from the text file (lossy %10.6f, like the reference) and from the state file
        g.set_fields({k: np.zeros(g.shape) for k in pkg.FIELDS})
        assert g.read_state(str(tmp_path / "a.bin")) == 2.5e-9
        got = g.fields()
        for k in f1:
            assert np.array_equal(got[k], f1[k]), k
        t = g.read_data(str(tmp_path / "a_end.dat"))
        assert abs(t - 2.5e-9) < 1e-6
        txt = g.fields()
    with pkg.Solver(p) as s:
        s.read_data(str(tmp_path / "b_end.dat"))
        one = s.fields()
        assert s.read_state(str(tmp_path / "b.bin")) == 2.5e-9
        back = s.fields()
    for k in f1:
        assert np.array_equal(txt[k], one[k]), k
        assert np.array_equal(back[k], f1[k]), k


def test_group_converged_start_equals_single_context(pkg, O):
    shape = (16, 8, 160)  # taller than the ~180 planes? no: 160 planes still converge with the reduced damping
    p = pkg.default_params(*shape)
    with pkg.Solver(p) as s:
        n1, r1 = s.initialization_converged(1e-8, 4000)
        f1 = s.fields()
    with pkg.Group(p, 4, devices=[0] * 4) as g:
        n4, r4 = g.initialization_converged(1e-8, 4000)
        f4 = g.fields()
    assert n1 == n4 and 0 < n1 < 4000
    err = O.rel_l2(f4, f1, {k: v for k, v in O.GROUPS.items() if k != "u"})
    assert max(err.values()) < 1e-10, err


def test_cpp_driver_with_slabs_writes_the_single_gpu_files(pkg, tmp_path):
    """ekpnp_main --gpus N (csrc/ekpnp_main.cpp over ekpnp_group_*): main.cu's whole sequence -
    initialization, time loop with Tecplot zones / current / umax lines, data_end.dat - on 4 z
    slabs gives the files of the one-GPU run.  The slabs solve the z-tridiagonal system in a
    different association (interface system), so the comparison is numeric, at the precision of
    the text formats (%g: 6 significant digits, %10.6f)."""
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ek-pnp-3d_amd", "ekpnp_main")
    if not os.path.exists(exe):
        pytest.skip("ekpnp_main not built")
    geo = ["--nx", "24", "--ny", "6", "--nz", "32", "--steps", "40", "--nsave", "15", "--print-current", "10"]
    outs = {}
    # (the four-slab run also takes the transport's knobs from the command line: --tune, round 5 - same bits under every setting)
    for tag, extra in (("one", []), ("four", ["--devices", "0,0,0,0", "--tune", "edge_chunks=2", "--tune", "lead_planes=0"]), ("rccl1", ["--devices", "0", "--transport", "rccl"])):
        d = tmp_path / tag
        d.mkdir()
        r = subprocess.run([exe, *geo, "--out", str(d), "--binary-state", "1", *extra], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (tag, r.stderr[-2000:])
        if tag == "four":
            assert "z slabs: 4, halo transport: device copies" in r.stdout
        if tag == "rccl1":
            assert "z slabs: 1, halo transport: RCCL" in r.stdout
        outs[tag] = (d, [float(l.split("Current = ")[1]) for l in r.stdout.splitlines() if "Current = " in l])
    one, cur1 = outs["one"]
    assert len(cur1) == 4
    for tag in ("four", "rccl1"):
        d, cur = outs[tag]
        assert np.allclose(cur, cur1, rtol=2e-5), (tag, cur, cur1)
        a, b = np.loadtxt(one / "data_end.dat"), np.loadtxt(d / "data_end.dat")
        assert a.shape == b.shape == (24 * 6 * 32, 12) and np.abs(a - b).max() <= 2e-6
        ua, ub = np.loadtxt(one / "umax.dat"), np.loadtxt(d / "umax.dat")
        assert ua.shape == ub.shape and np.abs(ua - ub).max() <= 2e-6
        la, lb = (one / "data.dat").read_text().splitlines(), (d / "data.dat").read_text().splitlines()
        assert len(la) == len(lb) and [l for l in la if not l[:1].isdigit()] == [l for l in lb if not l[:1].isdigit()]  # headers, ZONE lines
        sa, sb = np.fromfile(one / "data_end.bin", dtype=np.float64, offset=40), np.fromfile(d / "data_end.bin", dtype=np.float64, offset=40)
        n = 24 * 6 * 32
        for i, k in enumerate(pkg.FIELDS):  # the lossless file: rounding-level agreement field by field
            x, y = sa[i * n:(i + 1) * n], sb[i * n:(i + 1) * n]
            assert np.abs(x - y).max() <= 1e-9 * (np.abs(x).max() + 1e-300) + (1e-11 if k.startswith("u") else 0.0), (tag, k)
    # restart of the slab run from its own whole-lattice files
    d = outs["four"][0]
    r = subprocess.run([exe, "--nx", "24", "--ny", "6", "--nz", "32", "--steps", "2", "--devices", "0,0,0,0", "--read-previous", "2", "--out", str(d)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "Reading previous data (binary)" in r.stdout, r.stderr[-2000:]
    r = subprocess.run([exe, "--nx", "24", "--ny", "6", "--nz", "32", "--steps", "2", "--devices", "0,0,0,0", "--read-previous", "1", "--out", str(d)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "Reading previous data..." in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("extra", [[], ["--devices", "0,0,0"]])
def test_ekpnp_main_batch_loop_writes_the_same_files(tmp_path, extra):
    """ekpnp_main --batch 1: ONE ekpnp_step(n) / ekpnp_group_step(n) call from each output mark to the next, with the knob
    batch_moments on (only a call's last step stores rho, u, c, cn, T) - the way a host gets the 56 B/node back that bench.py
    reports beside its headline.  Every file of the run (data.dat with its three zones, umax.dat, data_end.dat, the lossless
    data_end.bin) and every `Current =` line equals the default loop's, which mirrors main.cu:189-224 call by call - byte for
    byte, on one context and on three slabs."""
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ek-pnp-3d_amd", "ekpnp_main")
    if not os.path.exists(exe):
        pytest.skip("ekpnp_main not built")
    geo = ["--nx", "40", "--ny", "6", "--nz", "33", "--steps", "47", "--nsave", "15", "--print-current", "10", "--uw", "3e-4", "--binary-state", "1"]
    runs = {}
    for tag, opt in (("loop", []), ("batch", ["--batch", "1"])):
        d = tmp_path / tag
        d.mkdir()
        r = subprocess.run([exe, *geo, *extra, *opt, "--out", str(d)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (tag, r.stderr[-2000:])
        runs[tag] = (d, [l for l in r.stdout.splitlines() if l.startswith("Iteration:")])
    (da, la), (db, lb) = runs["loop"], runs["batch"]
    assert la == lb and len([l for l in la if "Current = " in l]) == 5
    for f in ("data.dat", "umax.dat", "data_end.dat", "data_end.bin"):
        a, b = (da / f).read_bytes(), (db / f).read_bytes()
        assert len(a) > 0 and a == b, f


# ---- BASELINE cfg4 / cfg5 at their full per-rank shapes (size-independent properties) -------------

def _uniform_profiles(run, p, steps):
    """x-y uniform PB start from the product, `steps` steps; z profiles at (0,0) and the far corner"""
    import bench

    prof, _ = bench.pb_profile_from_product(_uniform_profiles.pkg, p)
    bench.product_pb_state(run, p, prof)
    run.fast_Poisson()
    run.init_equilibrium()
    run.step(steps)
    out = {}
    for k in ("rho", "c", "cn", "phi", "T", "Ez", "uz", "ux"):
        v = run.get_field(k)
        assert np.isfinite(v).all(), k
        assert np.abs(v[:, -1, -1] - v[:, 0, 0]).max() <= 1e-12 * np.abs(v).max(), k  # uniform at the far corner of the index space too
        out[k] = v[:, 0, 0].copy()
    return out


def _same_profiles(a, b):
    for k in a:
        tol = 1e-7 if k in ("uz", "ux") else 1e-11
        assert np.abs(a[k] - b[k]).max() <= tol * np.abs(b[k]).max(), (k, np.abs(a[k] - b[k]).max(), np.abs(b[k]).max())


def test_cfg4_whole_lattice_512x512x1024_in_place_periodic_tiles(pkg, O):
    """cfg4's lattice (268 M nodes x 4 lattices, 276 GB in place: population element indices beyond 2^32, where the
    reference's 32-bit index arithmetic, LBM.cu:27-30, has long wrapped) on one MI355X, with x-y structured data: every one
    of the 8 x 8 replicas of a perturbed 64 x 64 tile - the one holding element 2^32 of a population buffer (plane 605,
    row 417) included - must equal the 64 x 64 x 1024 run (tests/_periodic_tiles.py)."""
    from _periodic_tiles import periodic_tile_check

    periodic_tile_check(pkg, O, 1024, 1, 3, 282e9, 32)


def test_cfg5_per_rank_slab_1024x1024x128_through_the_transport(pkg):
    """cfg5's per-rank shape (1024x1024 planes, 128 of them: 302 MB of halo per direction and
    step) on the multi-rank code path: a slab context with the library's RCCL transport attached,
    one rank, the ring closing on itself.  Same property: equal to a 64x64x128 single-context run."""
    import torch

    _uniform_profiles.pkg = pkg
    free_b, _ = torch.cuda.mem_get_info()
    if free_b < 255e9:
        pytest.skip("needs 250 GB of free HBM")
    p = pkg.default_params(64, 64, 128)
    with pkg.Solver(p) as s:
        small = _uniform_profiles(s, p, 3)
    p = pkg.default_params(1024, 1024, 128)
    s = pkg.Solver(p, 0, 1, slab=True)
    try:
        s.attach_comm(pkg.comm_unique_id())
        big = _uniform_profiles(s, p, 3)
        assert s.device_bytes() > 245e9
    finally:
        s.close()
    _same_profiles(big, small)


def test_group_checkpoint_continues_and_is_interchangeable_with_one_context(pkg, O, tmp_path):
    shape = (24, 6, 48)
    po = O.default_params(*shape)
    po.pb_iterations = 8
    p = pkg.default_params(*shape)
    p.pb_iterations = 8
    ck = str(tmp_path / "g.ck")
    with pkg.Group(p, 4, devices=[0] * 4) as g:
        g.initialization()
        g.set_fields(O.perturb_fields(po, g.fields()))
        g.fast_Poisson(); g.init_equilibrium(); g.step(5)
        g.save_checkpoint(ck)
        g.step(4)
        want = g.fields()
    with pkg.Group(p, 4, devices=[0] * 4) as g:   # same decomposition: bit for bit
        assert abs(g.load_checkpoint(ck) - 5 * p.dt) < 1e-22
        g.step(4)
        got = g.fields()
    for k in want:
        assert np.array_equal(got[k], want[k]), k
    others = []
    with pkg.Group(p, 2, devices=[0] * 2) as g:   # another decomposition, and one context:
        g.load_checkpoint(ck); g.step(4)           # same run up to the association of the z solve
        others.append(g.fields())
    with pkg.Solver(p) as s:
        s.load_checkpoint(ck); s.step(4)
        others.append(s.fields())
        s.save_checkpoint(str(tmp_path / "s.ck"))
    for f in others:
        err = O.rel_l2(f, want)
        assert all(v < (1e-7 if k == "u" else 1e-11) for k, v in err.items()), err
    with pkg.Group(p, 3, devices=[0] * 3) as g:   # a single context's file into a group
        g.load_checkpoint(str(tmp_path / "s.ck")); g.step(2)
        a = g.fields()
    with pkg.Solver(p) as s:
        s.load_checkpoint(str(tmp_path / "s.ck")); s.step(2)
        b = s.fields()
    err = O.rel_l2(a, b)
    assert all(v < (1e-7 if k == "u" else 1e-11) for k, v in err.items()), err


@pytest.mark.parametrize("nl", [3, 1])
def test_group_with_fewer_lattices_vs_oracle(pkg, O, nl):
    """cfg2 physics (f + h + hn, Ra = 0) and cfg1 physics (fluid only, body force) on 3 slabs."""
    po = O.default_params(40, 6, 36)
    po.pb_iterations = 8
    po.Ra = 0.0
    if nl == 1:
        po.chargeinf, po.TH, po.exf = 0.0, 0.0, 1e9
    po.n_lattices = nl
    orc = O.Oracle(po)
    orc.initialization()
    start = O.perturb_fields(po, orc.fields()) if nl == 3 else orc.fields()
    orc.set_fields(start); orc.fast_poisson(); orc.init_equilibrium(); orc.step(6)
    want = orc.fields()
    with pkg.Group(_mirror(pkg, po), 3, devices=[0] * 3) as g:
        g.initialization()
        g.set_fields(start); g.fast_Poisson(); g.init_equilibrium(); g.step(6)
        got = g.fields()
    skip = ("T",) if nl == 3 else ("T", "c", "cn", "phi", "E")
    groups = {k: v for k, v in O.GROUPS.items() if k not in skip}
    _check(O, got, want, groups, f"{nl} lattices")


@pytest.mark.parametrize("shape,nslabs,in_place", [((24, 6, 51), 2, 0), ((24, 6, 52), 3, 1), ((70, 5, 65), 8, 0), ((16, 8, 33), 5, 0), ((20, 6, 131), 7, 1)])
def test_uneven_slabs_vs_oracle(pkg, O, shape, nslabs, in_place):
    """NZ not divisible by the number of slabs (the reference's usual NZ = 2^k + 1, LBM.h:35): the
    slabs differ by one plane; every rank solves the interface system with each block's own
    (u_1, u_m)."""
    po = O.default_params(*shape)
    po.pb_iterations = 10
    ref = _oracle_run(O, po, 6)
    p = _mirror(pkg, po)
    p.in_place = in_place
    with pkg.Group(p, nslabs, devices=[0] * nslabs) as g:
        ext = [g.slab_extent(i) for i in range(nslabs)]
        assert ext[0][0] == 0 and ext[-1][0] + ext[-1][1] == shape[2]
        assert all(a[0] + a[1] == b[0] for a, b in zip(ext, ext[1:])) and max(e[1] for e in ext) - min(e[1] for e in ext) == 1
        _drive(O, g, po, ref, 6)


def test_attached_slab_collectives_and_file_turns(pkg, O, tmp_path):
    """The multi-process routes of an attached slab - ncclAllReduce for current / umax / the converged
    start's residual, the rank-by-rank turns of the whole-lattice writers and of read_data - forced
    on one rank (EKPNP_TEAM_FORCE_COLLECTIVES, read when the communicator is attached): results and
    files must be the single-context ones."""
    import subprocess
    import sys

    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as G
pkg = G.load_package(); O = G.load_oracle()
out = sys.argv[1]
shape = (20, 6, 24)
p = pkg.default_params(*shape); p.pb_iterations = 8
po = O.default_params(*shape)
def drive(s, tag):
    n, res = s.initialization_converged(1e-9, 3000)
    s.set_fields(O.perturb_fields(po, s.fields()))
    s.fast_Poisson(); s.init_equilibrium(); s.step(5)
    s.save_data_tecplot(f"{out}/{tag}.dat", 1e-9, first=True)
    s.save_data_end(f"{out}/{tag}_end.dat", 1e-9)
    s.record_umax(f"{out}/{tag}_umax.dat", 1e-9, append=False)
    cur, um = s.current(), s.umax()
    t = s.read_data(f"{out}/{tag}_end.dat")
    return n, cur, um, s.fields()
with pkg.Solver(p) as s:
    a = drive(s, "one")
s = pkg.Solver(p, 0, 1, slab=True)
s.attach_comm(pkg.comm_unique_id())
b = drive(s, "slab")
s.close()
assert a[0] == b[0], (a[0], b[0])
assert abs(a[1] - b[1]) <= 1e-9 * abs(a[1]) and abs(a[2] - b[2]) <= 1e-6 * abs(a[2]) + 1e-30, (a[1:3], b[1:3])
for k in a[3]:
    # what read_data brought back is %%10.6f text of fields that agree to a few ulps (the slab solves z in another elimination
    # order): identical up to ONE unit of the last printed digit where a value sits on a rounding boundary
    assert np.abs(a[3][k] - b[3][k]).max() <= 1.000001e-6, (k, float(np.abs(a[3][k] - b[3][k]).max()))
print("OK")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EKPNP_TEAM_FORCE_COLLECTIVES="1")
    r = subprocess.run([sys.executable, "-c", code, str(tmp_path)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
    for name in ("umax.dat",):
        assert (tmp_path / f"one_{name}").read_bytes() == (tmp_path / f"slab_{name}").read_bytes()
    a, b = np.loadtxt(tmp_path / "one_end.dat"), np.loadtxt(tmp_path / "slab_end.dat")
    assert a.shape == b.shape and np.abs(a - b).max() <= 2e-6


def test_interior_rank_at_cfg5_width_vs_oracle(pkg, O):
    """cfg5's planes (1024 x 1024: rows of 16 tiles, 520 half-spectrum columns) in THREE slabs of 4 planes: the middle slab
    owns no plate - both its z neighbours are other slabs, every row of its z block couples through the interface
    system, its phi / E kernel takes both halo planes - which is what 6 of the 8 ranks of cfg5 look like.  Against the
    oracle: Poisson solve, two steps, diagnostics."""
    po = O.default_params(1024, 1024, 12)
    po.pb_iterations = 2
    orc = O.Oracle(po)
    try:
        orc.initialization()
        start = O.perturb_fields(po, orc.fields())
        orc.set_fields(start)
        orc.fast_poisson()
        pois = {k: orc.field(k).copy() for k in ("phi", "Ex", "Ey", "Ez")}
        orc.init_equilibrium()
        orc.step(2)
        want, cur, um = orc.fields(), orc.current(), orc.umax()
    finally:
        orc.close()
    with pkg.Group(_mirror(pkg, po), 3, devices=[0, 0, 0]) as g:
        assert [g.slab_extent(i) for i in range(3)] == [(0, 4), (4, 4), (8, 4)]
        g.set_fields(start)
        g.fast_Poisson()
        _check(O, {k: g.get_field(k) for k in pois}, pois, {"phi": ["phi"], "E": ["Ex", "Ey", "Ez"]}, "poisson")
        g.init_equilibrium()
        g.step(2)
        _check(O, g.fields(), want, where="step 2")
        assert abs(g.current() - cur) <= 1e-8 * abs(cur)
        assert abs(g.umax() - um) <= 1e-6 * abs(um) + 1e-30


@pytest.mark.parametrize("shape", [(512, 512, 192), (1024, 1024, 32)])
def test_cfg4_planes_decomposed_over_8_slabs_equal_one_context(pkg, O, shape):
    """cfg4's full 512 x 512 planes as a DECOMPOSITION over 8 slabs: 512 x 512 x 192 in 8 in-place slabs of 24 planes next
    to each other on the one GPU (rounds 2-3: 768 planes, round 4: 384; the far end of the index space is since round 4 the
    business of the periodic-tile tests, which see x-y structure at cfg3's and cfg4's full heights - this test keeps the
    8-way decomposition at full plane width; halved again in round 5 to keep the suite near 500 s: the same kernel
    instantiations run, the slab z solve's short-column form included) against the same lattice in ONE in-place context -
    same perturbed x-y-z dependent start, 5 steps.  The single context itself is pinned to the oracle at this width by
    test_full_size_vs_oracle[cfg3_width].
    And cfg5's 1024 x 1024 planes over EIGHT slabs (six of them interior, as in cfg5@8) at the smallest height a slab
    allows: 1024 x 1024 x 32 in 8 in-place slabs of 4 planes (own plane transforms, four modes per wavefront in the z
    solve, 302 MB halo messages) against one context."""
    import importlib.util
    import torch

    if torch.cuda.mem_get_info()[0] < 150 * 10**9:
        pytest.skip("needs 150 GB of free device memory")
    spec = importlib.util.spec_from_file_location("group_overhead", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "group_overhead.py"))
    go = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(go)
    p = pkg.default_params(*shape)
    p.in_place = 1
    keys = ("rho", "c", "cn", "phi", "T", "ux", "uy", "uz", "Ez")
    with pkg.Solver(p) as s:
        go.start(s, p, s.shape)
        s.step(5)
        want = {k: s.get_field(k) for k in keys}
        cur = s.current()
    with pkg.Group(p, 8, devices=[0] * 8) as g:
        go.start(g, p, g.shape)
        g.step(5)
        for k in keys:
            got = g.get_field(k)
            err = float(np.sqrt(((got - want[k]) ** 2).sum() / (want[k] ** 2).sum()))
            assert err <= (1e-7 if k in ("ux", "uy", "uz") else 1e-9), (k, err)
            del got
        assert abs(g.current() - cur) <= 1e-8 * abs(cur)


@pytest.mark.parametrize("shape,nslabs", [((70, 5, 32), 4), ((130, 4, 64), 2)])
def test_serial_slab_z_solve_still_matches_the_oracle(pkg, O, monkeypatch, shape, nslabs):
    """Slabs of more than 512 unknown rows (cfg4's 1024 planes on one or two devices in place) and EKPNP_TRI_PARTITION=0 keep
    the serial pair k_slab_thomas_local + k_slab_reduce_correct of round 2; every other slab test now takes the read-once
    pair, so this one pins the fallback."""
    monkeypatch.setenv("EKPNP_TRI_PARTITION", "0")
    po = O.default_params(*shape)
    po.pb_iterations = 15
    ref = _oracle_run(O, po, 7)
    with pkg.Group(_mirror(pkg, po), nslabs, devices=[0] * nslabs) as g:
        _drive(O, g, po, ref, 7)


def test_a_failed_group_verb_returns_drained_and_poisons_the_group(tmp_path):
    """a per-slab failure inside an in-process group is local (one host thread drives all slabs).  The
    library's fault-injection knob rejects ONE launch on ONE slab - the second bulk sweep of slab 1 of 3, i.e. in the
    middle of ekpnp_group_step(4), after slab 0 has already collided and begun nothing it waits for.  The test must see:
    the error returned from ekpnp_group_step with the kernel and the slab named; nothing of the group still running on
    the device when the call returns (a hipDeviceSynchronize right after returns at once and cleanly); every further verb
    refused with the FIRST failure; ekpnp_group_destroy clean; and the device fully usable afterwards in the same process
    (a fresh group runs the same steps and matches a run that never failed) - no leaked GPU holder."""
    import subprocess
    import sys

    code = r'''
import sys, time, ctypes, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as G
pkg = G.load_package()
import torch
p = pkg.default_params(40, 6, 36); p.pb_iterations = 4
free0 = torch.cuda.mem_get_info()[0]
g = pkg.Group(p, 3, devices=[0, 0, 0])
g.initialization(); g.init_equilibrium()
try:
    g.step(4)
    print("NOERR")
except pkg.EkpnpError as e:
    print("ERR1", e)
hip = ctypes.CDLL("libamdhip64.so")
t0 = time.perf_counter(); rc = hip.hipDeviceSynchronize(); dt = time.perf_counter() - t0
print("SYNC", rc, "FAST" if dt < 0.5 else "SLOW %%.2f" %% dt)
for verb in (lambda: g.step(1), lambda: g.get_field("rho"), lambda: g.fast_Poisson(), lambda: g.synchronize()):
    try:
        verb(); print("NOT REFUSED")
    except pkg.EkpnpError as e:
        print("ERR2", e)
g.close()
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
# the device is fully usable: a fresh group in the same process runs (one step stays below the injected launch count) ...
g2 = pkg.Group(p, 3, devices=[0, 0, 0])
g2.initialization(); g2.init_equilibrium(); g2.step(1)
print("FRESH", bool(np.isfinite(g2.get_field("rho")).all()))
g2.close()
torch.cuda.synchronize()
free2 = torch.cuda.mem_get_info()[0]
# ... and gives back exactly what it took: the poisoned group had released everything (the first group of a process also
# leaves rocFFT / runtime caches behind, which is why the baseline is taken after it)
print("LEAK", abs(free1 - free2) > (8 << 20), free0 - free1, free1 - free2)
'''
    body = code % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EKPNP_INJECT_LAUNCH_FAILURE="k_collide_bulk@1#4")  # slab 1 sweeps its interior in 2 launches per step (lead-in + rest): the 4th is the middle of step 2
    r = subprocess.run([sys.executable, "-c", body], env=env, capture_output=True, text=True, timeout=300)
    out = r.stdout
    assert r.returncode == 0, (out, r.stderr[-3000:])
    assert "NOERR" not in out and "ERR1" in out and "kernel k_collide_bulk" in out and "slab 1" in out, out
    assert "SYNC 0 FAST" in out, out
    assert out.count("ERR2") == 4 and "NOT REFUSED" not in out and out.count("poisoned") >= 4, out
    for line in out.splitlines():
        if line.startswith("ERR2"):
            assert "kernel k_collide_bulk" in line, line  # the first failure, kept
    assert "FRESH True" in out and "LEAK False" in out, out


@pytest.mark.parametrize("shape,nslabs,in_place,nl", [((70, 5, 45), 3, 0, 4), ((130, 4, 33), 2, 1, 4), ((24, 6, 29), 5, 0, 3), ((16, 6, 17), 1, 0, 4)])
def test_edge_planes_without_pack_and_unpack_are_bitwise_the_copied_halos(pkg, O, monkeypatch, shape, nslabs, in_place, nl):
    """Round 4: the launches that collide a slab's first and last plane store their 9 outgoing directions straight into
    the send buffers and pull their 9 incoming ones straight out of the receive buffers (k_collide_edge, the plates'
    k_collide_wall); k_halo_pack / k_halo_unpack and the ghost planes drop out of the step.  EKPNP_HALO_DIRECT=0 keeps the
    copies of rounds 1-3: every field must come out the same bit for bit - uneven slabs, in-place populations (staged
    edge planes), three lattices, rows that end inside a tile, and one slab whose ring closes on itself (both plates
    exchange with each other: the wall-to-wall ghost loop of gpu_stream, LBM.cu:1972,1975)."""
    p = pkg.default_params(*shape)
    p.pb_iterations = 6
    p.in_place = in_place
    p.n_lattices = nl
    if nl < 4:
        p.Ra = 0.0
    outs = []
    # direct halos with both faces in one launch (k_collide_faces, the default) against the copies of rounds 1-3; the
    # launch-per-face partner of the faces kernel (EKPNP_MERGED_FACES=0, read once per process) runs in the three-rank
    # case of tests/test_slab_gpu.py::test_native_rccl_ranks_sharing_one_gpu
    for direct in ("1", "0"):
        monkeypatch.setenv("EKPNP_HALO_DIRECT", direct)
        with pkg.Group(p, nslabs, devices=[0] * nslabs) as g:
            g.initialization()
            g.set_fields(O.perturb_fields(p, g.fields()))
            g.fast_Poisson(); g.init_equilibrium()
            g.step(7)
            g.init_equilibrium()  # restart from the fields: the first step after it does not pull
            g.step(4)
            outs.append(g.fields())
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), (k, float(np.abs(outs[0][k] - outs[1][k]).max()))


def test_slab_edge_values_with_charge_at_one_plate(pkg):
    """k_slab_edges truncates the dot products behind the interface values to the rows where the unit response
    u_j is not below 2^-66 u_1 - an ABSOLUTE bound (relative to max |rhs|), not one relative to the kept sum.  The worst
    case for it: charge confined to the two planes next to ONE plate (an electric double layer far thinner than a slab), so
    that the far edge of every slab sees right-hand sides that are huge at one end and zero elsewhere.  Four slabs of a
    258-plane channel against one context, tolerance relative to max |phi|; and the same with the charge at the upper plate."""
    shape = (64, 16, 258)
    p = pkg.default_params(*shape)
    rng = np.random.default_rng(23)
    for at_top in (False, True):
        cc = np.full(shape[::-1], 0.01)
        cn = np.full(shape[::-1], 0.01)
        planes = slice(-3, -1) if at_top else slice(1, 3)
        cc[planes] *= 1.0 + 50.0 * rng.random(cc[planes].shape)
        with pkg.Solver(p) as s:
            s.set_field("c", cc); s.set_field("cn", cn)
            s.fast_Poisson()
            one = {k: s.get_field(k) for k in ("phi", "Ez")}
        with pkg.Group(p, 4, devices=[0] * 4) as g:
            g.set_field("c", cc); g.set_field("cn", cn)
            g.fast_Poisson()
            four = {k: g.get_field(k) for k in ("phi", "Ez")}
        for k in one:
            err = np.abs(four[k] - one[k]).max() / np.abs(one[k]).max()
            assert err < 1e-12, (at_top, k, err)


def test_cpp_driver_ends_cleanly_when_one_slab_fails(tmp_path):
    """`ekpnp_main --devices 0,0,0` (one process, three slabs, no control plane) with the fault-injection knob rejecting a
    launch on slab 1 in the middle of the run: the program must END - non-zero, with the slab and the kernel named - rather
    than hang on the slabs that went ahead (round 4: a failing group verb drains every stream and poisons the group;
    `fail()` of csrc/ekpnp_main.cpp then destroys it)."""
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ek-pnp-3d_amd", "ekpnp_main")
    if not os.path.exists(exe):
        pytest.skip("ekpnp_main not built")
    env = dict(os.environ, EKPNP_INJECT_LAUNCH_FAILURE="k_collide_bulk@1#7")
    r = subprocess.run([exe, "--nx", "24", "--ny", "6", "--nz", "36", "--steps", "40", "--nsave", "15", "--print-current", "10",
                        "--out", str(tmp_path), "--devices", "0,0,0"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 1, (r.returncode, r.stderr[-2000:])
    assert "slab 1" in r.stderr and "kernel k_collide_bulk" in r.stderr, r.stderr[-2000:]


# ---- round 5: the EDGE all-gather in mode blocks ("edge_chunks"), the transport's knobs on a live team, stage times ----------

def _fields_bits(run):
    f = run.fields()
    return {k: np.ascontiguousarray(v).copy() for k, v in f.items()}


@pytest.mark.parametrize("shape,nslabs,in_place", [((48, 6, 24), 3, 0), ((130, 4, 64), 4, 1), ((16, 12, 16), 2, 0), ((512, 512, 24), 3, 0)])
def test_edge_chunks_give_the_same_bits_and_match_the_oracle(pkg, O, shape, nslabs, in_place):
    """the z coupling of the slab solve (the all-gather that replaces the z part of poisson.cu:86-92) in
    pipelined mode blocks.  Every mode goes through the same operations whatever the number of blocks, so ALL fields after a
    solve and after 5 steps are bit-identical to the one-block run; the one-block run is held against the oracle.
    Grids: nxh = 32 (4 column groups: 2, 3 and 5 -> 4 blocks), 72 (9 groups, uneven blocks, in place, serial z sweeps of the
    64-plane channel's 16-plane slabs), 16 (2 groups), and 512 x 512 planes where the library's OWN row / column passes run
    and the column pass is cut into the blocks as well (264 = 33 groups)."""
    po = O.default_params(*shape)
    po.pb_iterations = 6
    big = shape[0] >= 512
    p = _mirror(pkg, po)
    p.in_place = in_place
    want = None
    if not big:
        ref = _oracle_run(O, po, 5)
    for chunks in (1, 2, 3, 5):
        with pkg.Group(p, nslabs, devices=[0] * nslabs) as g:
            g.tune("edge_chunks", chunks)
            if not big:
                _drive(O, g, po, ref, 5)  # every chunk count against the oracle too
            else:
                g.initialization()
                g.set_fields(O.perturb_fields(po, g.fields()))
                g.fast_Poisson()
                g.init_equilibrium()
                g.step(3)
            got = _fields_bits(g)
        if want is None:
            want = got
        else:
            for k in want:
                assert np.array_equal(got[k], want[k]), (chunks, k, float(np.abs(got[k] - want[k]).max()))
    if big:  # the own-transform planes against a single context (same kernels unchunked), which the parity tests hold against the oracle
        with pkg.Solver(p) as s:
            s.initialization()
            s.set_fields(O.perturb_fields(po, s.fields()))
            s.fast_Poisson()
            s.init_equilibrium()
            s.step(3)
            _check(O, s.fields(), want, where="512 x 512 planes, slabs vs one context")


def test_edge_chunks_when_the_slabs_run_different_z_solve_kernels(pkg, O):
    """The mode blocks are the pieces of a COLLECTIVE: their boundaries must be the same on every rank, whatever z-solve kernel a
    rank runs.  48 x 6 x 388 in three slabs gives unknown-row counts of 128, 129, 129: slab 0 runs k_slab_part<8,16> (32 modes
    per workgroup), the others <8,32> (16 modes) - with block units derived from the slab's own kernel the ranks disagreed
    about the blocks (units of 2 and of 1 column groups at ny = 6).  Bit-identical fields for 1, 2, 3 blocks, and against the
    oracle (uniform start + perturbation: the reference's Picard start-up diverges on a channel this tall)."""
    po = O.default_params(48, 6, 388)
    orc = O.Oracle(po)
    try:
        orc.gpu_initialization()
        f0 = orc.fields()
        f0["c"] = np.full_like(f0["c"], po.chargeinf)  # (gpu_initialization leaves the ions to gpu_PBE: LBM.cu:111-146)
        f0["cn"] = np.full_like(f0["cn"], po.chargeinf)
        start = O.perturb_fields(po, f0)
        orc.set_fields(start)
        orc.fast_poisson()
        orc.init_equilibrium()
        orc.step(4)
        ref = orc.fields()
        assert np.abs(ref["Ex"]).max() > 1.0  # a real field, not rounding noise
    finally:
        orc.close()
    p = _mirror(pkg, po)
    want = None
    for chunks in (1, 2, 3):
        with pkg.Group(p, 3, devices=[0, 0, 0]) as g:
            assert [g.slab_extent(i)[1] for i in range(3)] == [129, 129, 130]
            g.tune("edge_chunks", chunks)
            g.set_fields(start)
            g.fast_Poisson()
            g.init_equilibrium()
            g.step(4)
            got = _fields_bits(g)
        _check(O, got, ref, where=f"{chunks} blocks")
        if want is None:
            want = got
        else:
            for k in want:
                assert np.array_equal(got[k], want[k]), (chunks, k)


def test_transport_knobs_on_a_live_one_rank_ring(pkg, O):
    """the knobs one GPU cannot decide are ekpnp_tune knobs of a live context now (bench.py's comm_ab runs
    a few steps under each after its timed region).  Here: the same lattice stepped under every setting gives the same bits,
    the stage times add up to the solve, an unknown knob is refused, and a context without a transport refuses the layout
    that only the library's transport can exchange."""
    po = O.default_params(48, 6, 24)
    po.pb_iterations = 8
    p = _mirror(pkg, po)

    def run(settings):
        s = pkg.Solver(p, 0, 1, slab=True)
        try:
            s.attach_comm(pkg.comm_unique_id())
            s.initialization()
            s.set_fields(O.perturb_fields(po, s.fields()))
            s.fast_Poisson()
            s.init_equilibrium()
            s.step(2)
            for k, v in settings:
                s.tune(k, v)
            s.kernel_timing(True)
            s.step(4)
            n, st = s.poisson_stage_timing_get()
            ns, tot = s.phase_timing_get()
            comm = s.comm_timing_get()  # (before timing is switched off: that forgets the exchanges bracketed so far)
            s.kernel_timing(False)
            assert n == ns == 4 and set(st) == set(pkg.STAGE_NAMES)
            assert all(v >= 0.0 for v in st.values()) and abs(sum(st.values()) - tot) <= 1e-3 * tot + 1e-3
            return _fields_bits(s), comm
        finally:
            s.close()

    want, comm0 = run([])
    assert comm0["edge"]["n"] == 4 and comm0["halo"]["n"] == 4 and comm0["phi"]["n"] == 4
    for settings in ([("inline_exchanges", 0)], [("comm_cus", 8)], [("comm_cus", 8), ("comm_cus", 0)], [("lead_planes", 0)], [("merged_faces", 0)],
                     [("edge_chunks", 4)], [("edge_p2p", 1)], [("edge_p2p", 1), ("edge_chunks", 2)],
                     [("inline_exchanges", 0), ("edge_chunks", 2), ("comm_cus", 16), ("lead_planes", 3)]):
        got, comm = run(settings)
        for k in want:
            assert np.array_equal(got[k], want[k]), (settings, k)
        if ("edge_chunks", 4) in settings:
            # four blocks asked for; nxh = 32 is 4 column groups, but ny = 6 and the z-solve kernel of a 22-row slab takes 32 modes
            # per workgroup: a block must hold whole workgroups (units of 2 groups) -> 2 blocks per solve, each bracketed
            assert comm["edge"]["n"] == 8
    s = pkg.Solver(p, 0, 1, slab=True)
    try:
        with pytest.raises(pkg.EkpnpError, match="unknown knob or bad value"):
            s.tune("edge_chunks", 4)  # no transport: the chunked layout has nobody to exchange it
        s.tune("edge_chunks", 1)
        s.attach_comm(pkg.comm_unique_id())
        with pytest.raises(pkg.EkpnpError, match="unknown"):
            s.tune("no_such_knob", 1)
        with pytest.raises(pkg.EkpnpError, match="bad value"):
            s.tune("comm_cus", 1000)
        s.tune("edge_chunks", 3)
        with pytest.raises(pkg.EkpnpError, match="edge_chunks"):
            s.call("poisson_stage1")  # the exported stages are for a caller's transport: one block only
    finally:
        s.close()


def test_batch_moments_on_slabs_changes_no_visible_bit(pkg, O):
    """"batch_moments" through the transport: a 3-slab group (uneven: 8, 8, 9 planes) and a one-rank RCCL ring step in one call
    of 6; the fields afterwards are the bits of the run that stores the moments in every step."""
    po = O.default_params(48, 6, 25)
    po.pb_iterations = 6
    p = _mirror(pkg, po)
    start = None
    out = {}
    for kind in ("group", "ring"):
        for knob in (0, 1):
            if kind == "group":
                run = pkg.Group(p, 3, devices=[0, 0, 0])
            else:
                run = pkg.Solver(p, 0, 1, slab=True)
                run.attach_comm(pkg.comm_unique_id())
            try:
                run.tune("batch_moments", knob)
                run.initialization()
                if start is None:
                    start = O.perturb_fields(po, run.fields())
                run.set_fields(start)
                run.fast_Poisson()
                run.init_equilibrium()
                run.step(6)
                out[kind, knob] = _fields_bits(run)
            finally:
                run.close()
        for k in out[kind, 0]:
            assert np.array_equal(out[kind, 0][k], out[kind, 1][k]), (kind, k)


def test_group_tune_and_stage_times(pkg, O):
    po = O.default_params(70, 5, 32)
    po.pb_iterations = 5
    p = _mirror(pkg, po)
    with pkg.Group(p, 4, devices=[0] * 4) as g:
        g.initialization()
        g.init_equilibrium()
        g.step(1)
        g.tune("edge_chunks", 2)
        g.tune("lead_planes", 0)
        with pytest.raises(pkg.EkpnpError):
            g.tune("edge_chunks", 99)
        g.step(2)
        assert np.isfinite(g.fields()["phi"]).all()
