"""CPU tests of the oracle: known-answer tests that are independent of the reference
 and - when present - the golden vectors produced by the reference's
own kernels on an MI355X (tests/golden/ref_*.npz, made by tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

from conftest import golden_path


def _ref_grid(O):
    p = O.default_params(50, 8, 51)
    # literal values of LBM.h:40-42
    p.Lx, p.Ly, p.Lz = 0.5e-6, 0.08e-6, 0.5e-6
    return p


def test_K1_poiseuille_profile(O):
    """Steady ux(z) = exf/(2 rho0 nu) (z - zw0)(zw1 - z) with half-way walls."""
    p = O.default_params(4, 4, 33)
    p.exf, p.chargeinf, p.Ra, p.TH, p.pb_iterations = 1e9, 0.0, 0.0, 0.0, 2
    o = O.Oracle(p)
    o.initialization()
    o.init_equilibrium()
    o.step(12000)
    ux = o.field("ux")[:, 0, 0]
    z = np.arange(p.nz) * p.dz
    ana = p.exf / (2 * p.rho0 * p.nu) * (z - 0.5 * p.dz) * ((p.nz - 1.5) * p.dz - z)
    rel = np.linalg.norm(ux[1:-1] - ana[1:-1]) / np.linalg.norm(ana[1:-1])
    assert rel < 1e-3  # 7.9e-4 at NZ=33 (Lambda = 1/12 wall slip), SURVEY.md K1
    assert abs(ux[0] + ux[1]) < 1e-12 * abs(ux[1])  # z==0 override: ux(0) = -ux(1)
    mass = o.population("f", 0).sum() + o.population("f", 1).sum()
    assert abs(mass / (p.rho0 * o.n) - 1) < 1e-12
    assert np.abs(o.field("uy")).max() < 1e-10 and np.abs(o.field("uz")).max() < 1e-10


def test_K2_poisson_equals_fft2_plus_tridiagonal(O):
    """The odd-extension 3-D DFT path == 2-D FFT in x,y + dense FD solve in z."""
    p = O.default_params(16, 12, 17)
    o = O.Oracle(p)
    o.gpu_initialization()
    rng = np.random.default_rng(0)
    o.field("c")[...] = 0.01 * (1 + 0.1 * rng.random(o.shape))
    o.field("cn")[...] = 0.01 * (1 + 0.1 * rng.random(o.shape))
    o.fast_poisson()
    phi = o.field("phi")
    nz = p.nz
    g = -p.convertCtoCharge * (o.field("c") - o.field("cn")) / p.eps
    rhs = g[1 : nz - 1].copy()
    rhs[0] -= p.voltage / p.dz**2
    rhs[-1] -= p.voltage2 / p.dz**2
    rh = np.fft.fft2(rhs, axes=(1, 2))
    kx = 2 * np.pi * np.fft.fftfreq(p.nx, d=p.Lx / p.nx)
    ky = 2 * np.pi * np.fft.fftfreq(p.ny, d=p.Ly / p.ny)
    m = nz - 2
    A = (np.diag(-2 * np.ones(m)) + np.diag(np.ones(m - 1), 1) + np.diag(np.ones(m - 1), -1)) / p.dz**2
    sol = np.zeros_like(rh)
    for j in range(p.ny):
        for i in range(p.nx):
            sol[:, j, i] = np.linalg.solve(A - (kx[i] ** 2 + ky[j] ** 2) * np.eye(m), rh[:, j, i])
    ref = np.fft.ifft2(sol, axes=(1, 2)).real
    assert np.abs(ref - phi[1 : nz - 1]).max() < 1e-13 * np.abs(phi).max()
    assert np.all(phi[0] == p.voltage) and np.all(phi[-1] == p.voltage2)


def test_K2b_poisson_eigenfunction(O):
    """sin modes are eigenfunctions with the eigenvalue of poisson.cu:176."""
    p = O.default_params(16, 8, 33)
    p.voltage = p.voltage2 = 0.0
    o = O.Oracle(p)
    o.gpu_initialization()
    z, y, x = np.meshgrid(np.arange(p.nz), np.arange(p.ny), np.arange(p.nx), indexing="ij")
    mx, my, mz = 2, 1, 3
    mode = np.sin(2 * np.pi * mx * x / p.nx) * np.cos(2 * np.pi * my * y / p.ny) * np.sin(np.pi * mz * z / (p.nz - 1))
    o.field("c")[...] = 1e-3 * mode
    o.field("cn")[...] = 0.0
    o.fast_poisson()
    kz = mz * 2 * np.pi / (2 * (p.nz - 1) * p.dz)
    mu = 4 / p.dz**2 * np.sin(kz * p.dz / 2) ** 2 + (2 * np.pi * mx / p.Lx) ** 2 + (2 * np.pi * my / p.Ly) ** 2
    want = p.convertCtoCharge * 1e-3 * mode / p.eps / mu
    assert np.abs(o.field("phi") - want).max() < 1e-12 * np.abs(want).max()


def test_K4_efield_is_central_difference(O):
    p = O.default_params(10, 6, 9)
    o = O.Oracle(p)
    rng = np.random.default_rng(1)
    phi = rng.standard_normal(o.shape)
    o.field("phi")[...] = phi
    o.efield()
    ex = 0.5 * (np.roll(phi, 1, 2) - np.roll(phi, -1, 2)) / p.dx
    ey = 0.5 * (np.roll(phi, 1, 1) - np.roll(phi, -1, 1)) / p.dy
    ez = 0.5 * (np.roll(phi, 1, 0) - np.roll(phi, -1, 0)) / p.dz
    ez[0], ez[-1] = ez[1], ez[-2]
    assert np.array_equal(o.field("Ex"), ex) and np.array_equal(o.field("Ey"), ey) and np.array_equal(o.field("Ez"), ez)


def test_K3_debye_layer_after_pb_init(O):
    """Mid-plane potential of the PB initial state vs the linearised Debye-Hueckel value."""
    p = _ref_grid(O)
    o = O.Oracle(p)
    o.initialization()
    lam = np.sqrt(p.eps * p.kB * p.roomT / p.electron / (2 * p.chargeinf * p.convertCtoCharge))
    assert abs(lam / p.dz - 9.2) < 0.1
    H = (p.nz - 1) * p.dz
    lin = p.voltage / np.cosh(H / (2 * lam))
    mid = o.field("phi")[p.nz // 2, 0, 0]
    assert abs(mid - lin) < 0.05 * abs(lin)
    c, cn = o.field("c"), o.field("cn")
    assert np.allclose(c * cn, p.chargeinf**2, rtol=1e-12)  # Boltzmann: c*cn = c_inf^2


def test_step_is_deterministic_and_finite(O):
    p = O.default_params(16, 12, 17)
    p.pb_iterations = 20
    outs = []
    for _ in range(2):
        o = O.Oracle(p)
        o.initialization()
        o.set_fields(O.perturb_fields(p, o.fields()))
        o.fast_poisson()
        o.init_equilibrium()
        o.step(5)
        outs.append(o.fields())
    for k in outs[0]:
        assert np.isfinite(outs[0][k]).all()
        assert np.array_equal(outs[0][k], outs[1][k])


def test_subkernels_compose_to_stream_collide_save(O):
    p = O.default_params(8, 6, 9)
    p.pb_iterations = 5
    a, b = O.Oracle(p), O.Oracle(p)
    for o in (a, b):
        o.initialization()
        o.set_fields(O.perturb_fields(p, o.fields()))
        o.fast_poisson()
        o.init_equilibrium()
    a.stream_collide_save()
    b.collide_save(); b.boundary(); b.stream(); b.bc_charge()
    for lat in ("f", "h", "hn", "temp"):
        for w in (0, 1):
            assert np.array_equal(a.population(lat, w), b.population(lat, w))


# ---- golden vectors from the reference's own kernels (made on an MI355X) --------------------
# tests/golden/ref_g*.npz hold outputs of the reference's LBM.cu / poisson.cu kernels built for
# gfx950 (oracle/build_ref.sh -> make_golden.py -> pack_golden.py).  The reference's Poisson
# solve leaks its FFT library's rounding residue into the DC mode (poisson.cu:177): one constant
# per solve on the interior phi, up to 4e-4 on a 5e-3 field.  The fixtures carry the measured
# constant of every solve (`*_shifts`); replaying the run with them injected must reproduce the
# reference to FP64 rounding.  TOL_U: see tests/test_parity_gpu.py.
TOL, TOL_U = 1e-12, 1e-7


def _need(name):
    path = golden_path(name)
    if not os.path.exists(path):
        pytest.skip(f"{name} missing (tests/golden/make_golden.py on the GPU box, then pack_golden.py)")
    return np.load(path)


def _check(err, where):
    bad = {k: v for k, v in err.items() if not v <= (TOL_U if k == "u" else TOL)}
    assert not bad, (where, err)


def test_golden_G1_default_run_init_and_100_steps(O):
    """The reference's own default run: initialization() with all 501 PB sweeps, then 100 steps."""
    g = _need("ref_g1.npz")
    p = _ref_grid(O)
    o = O.Oracle(p)
    o.initialization_shifts(g["init_shifts"])
    f = o.fields()
    for k, v in f.items():
        if k in ("Ex", "Ey", "ux", "uy", "uz"):
            continue  # zero up to the rounding noise of the oracle's DFT (1e-10 V/m next to Ez = 5e4)
        # the reference's fields are bit-exactly x-y uniform; the oracle's DFT leaves rounding noise
        assert np.abs(v - v[:, :1, :1]).max() <= 1e-12 * (np.abs(v).max() + 1e-300), f"{k} must stay x-y uniform"
    col = lambda d: {k: v[:, 0, 0] for k, v in d.items()}  # noqa: E731
    groups = {k: v for k, v in O.GROUPS.items() if k != "u"}  # u == 0 exactly after initialization
    _check(O.rel_l2(col(f), {k: g["init_" + k] for k in O.FIELDS}, groups), "init")
    assert not np.any(f["ux"]) and not np.any(g["init_ux"])
    o.init_equilibrium()
    done = 0
    for mark in (int(m) for m in g["marks"]):
        o.step_shifts(g["step_shifts"][done:mark])
        done = mark
        _check(O.rel_l2(col(o.fields()), {k: g[f"step{mark}_{k}"] for k in O.FIELDS}), f"step {mark}")


def test_golden_G1_dc_leak_is_what_separates_exact_from_reference(O):
    """Without the measured shifts the exact solve differs from the reference's run by the leak
    (this documents why 'DC = 0' is the canonical result)."""
    g = _need("ref_g1.npz")
    assert np.abs(g["init_shifts"]).max() > 1e-4  # 4e-4 on a 5.3e-3 potential
    residues = -g["init_shifts"] * (50 * 8 * 100)  # shift = -residue / size, LBM.h:38
    assert np.abs(residues - np.round(residues * 128) / 128).max() < 1e-6  # multiples of ulp(5e13) = 2^-7


def test_golden_G2_perturbed_3d_run(O):
    g = _need("ref_g2.npz")
    p = _ref_grid(O)
    ys = list(g["ysel"])
    o = O.Oracle(p)
    o.gpu_initialization()
    o.set_fields({k: g["input_" + k] for k in ("rho", "c", "cn", "T", "ux", "uy", "uz")})
    sub = lambda d: {k: v[:, ys, :] for k, v in d.items()}  # noqa: E731
    o.fast_poisson(float(g["shifts"][0]))
    _check(O.rel_l2(sub(o.fields()), {k: g["step0_" + k] for k in O.FIELDS}, {"phi": ["phi"], "E": ["Ex", "Ey", "Ez"]}), "step 0")
    o.init_equilibrium()
    done = 0
    for mark in (int(m) for m in g["marks"]):
        o.step_shifts(g["shifts"][1 + done : 1 + mark])
        done = mark
        _check(O.rel_l2(sub(o.fields()), {k: g[f"step{mark}_{k}"] for k in O.FIELDS}), f"step {mark}")


def test_golden_G3_body_force_channel(O):
    """exf = 1e9, chargeinf = 0, Ra = 0, TH = 0: rho and u do not see phi, no shift involved."""
    g = _need("ref_g3.npz")
    p = _ref_grid(O)
    p.exf, p.chargeinf, p.Ra, p.TH = 1e9, 0.0, 0.0, 0.0
    o = O.Oracle(p)
    o.initialization()
    o.init_equilibrium()
    done = 0
    full = os.environ.get("EKPNP_LONG_TESTS") == "1"  # the 3000-step mark takes minutes on one core
    for mark in (int(m) for m in g["marks"] if full or m <= 100):
        o.step(mark - done)
        done = mark
        f = {k: o.field(k)[:, [0, 3, 5], :] for k in ("rho", "ux", "uy", "uz")}
        err = O.rel_l2(f, {k: g[f"step{mark}_{k}"] for k in f}, {"rho": ["rho"], "u": ["ux", "uy", "uz"]})
        assert err["rho"] < 1e-12 and err["u"] < 1e-9, (mark, err)


def test_golden_G7_moving_wall(O):
    """uw = 1e-3 on the upper plate (LBM.cu:1896-1927, including "+ multis" on direction 3 only),
    chargeinf = 0, Ra = 0, TH = 0: rho and u are free of the DC leak."""
    g = _need("ref_g7.npz")
    p = _ref_grid(O)
    p.uw, p.chargeinf, p.Ra, p.TH = 1e-3, 0.0, 0.0, 0.0
    o = O.Oracle(p)
    o.initialization()
    o.init_equilibrium()
    full = os.environ.get("EKPNP_LONG_TESTS") == "1"
    done = 0
    for mark in (int(m) for m in g["marks"] if full or m <= 100):
        o.step(mark - done)
        done = mark
        f = {k: o.field(k)[:, [0, 3, 5], :] for k in ("rho", "ux", "uy", "uz")}
        err = O.rel_l2(f, {k: g[f"step{mark}_{k}"] for k in f}, {"rho": ["rho"], "u": ["ux", "uy", "uz"]})
        assert err["rho"] < 1e-12 and err["u"] < 1e-8, (mark, err)
    assert np.abs(g["step100_uy"]).max() > 0  # the direction-3 quirk drives a y velocity at the wall


def test_golden_G5_poisson_alone(O):
    g = _need("ref_g5.npz")
    p = _ref_grid(O)
    ys = list(g["ysel"])
    o = O.Oracle(p)
    o.gpu_initialization()
    o.set_fields({"c": g["input_c"], "cn": g["input_cn"]})
    o.fast_poisson(float(g["shift"]))
    got = {k: o.field(k)[:, ys, :] for k in ("phi", "Ex", "Ey", "Ez")}
    _check(O.rel_l2(got, {k: g["out_" + k] for k in got}, {"phi": ["phi"], "E": ["Ex", "Ey", "Ez"]}), "poisson")
    # and the exact solve differs from the reference by exactly that constant on the interior
    o.fast_poisson(0.0)
    d = g["out_phi"][1:-1] - o.field("phi")[1:-1][:, ys, :]
    assert np.abs(d - float(g["shift"])).max() < 1e-16


# ---- diagnostics ---------------------------------------------------

def test_current_and_umax_restate_the_reference_loops(O):
    p = O.default_params(12, 7, 9)
    o = O.Oracle(p)
    rng = np.random.default_rng(11)
    c, cn, ez, uz = (rng.random(o.shape) for _ in range(4))
    o.set_fields({"c": c, "cn": cn, "Ez": ez, "uz": uz - 0.5})
    ce = 2 * c[-2] - c[-3]
    cne = 2 * cn[-2] - cn[-3]
    want = ((ce - cne) * ez[-1]).sum() * p.K * p.dz * p.dz  # LBM.cu:2689-2708
    assert abs(o.current() - want) <= 1e-13 * abs(want)
    assert o.umax() == max(0.0, (uz - 0.5).max())  # LBM.cu:2718,2744
    o.set_fields({"uz": -uz})
    assert o.umax() == 0.0
    assert np.array_equal(o.field("c"), c)  # current() works on host copies in the reference


def test_golden_current_of_the_reference_runs(O):
    """double current(c, cn, ez) evaluated by the reference itself at the marks of G1/G2/G6."""
    g2 = _need("ref_g2.npz")
    if "current" not in g2.files:
        pytest.skip("fixtures predate the current() golden")
    p = _ref_grid(O)
    o = O.Oracle(p)
    for i, m in enumerate(int(v) for v in g2["marks"]):
        c = np.zeros(o.shape); cn = np.zeros(o.shape); ez = np.zeros(o.shape)
        c[-3:], cn[-3:], ez[-1] = g2[f"cur{m}_c"], g2[f"cur{m}_cn"], g2[f"cur{m}_Ez"]
        o.set_fields({"c": c, "cn": cn, "Ez": ez})
        want = float(g2["current"][i])
        assert abs(o.current() - want) <= 1e-14 * abs(want), (m, o.current(), want)
    g1 = _need("ref_g1.npz")
    for i, m in enumerate(int(v) for v in g1["marks"]):
        bc = lambda k: np.broadcast_to(g1[f"step{m}_{k}"][:, None, None], o.shape)  # noqa: E731
        o.set_fields({"c": bc("c"), "cn": bc("cn"), "Ez": bc("Ez")})
        want = float(g1["current"][i])
        assert abs(o.current() - want) <= 1e-13 * abs(want), (m, o.current(), want)


def test_golden_G4_per_kernel_vectors(O):
    """the four launches of stream_collide_save (LBM.cu:474-477) one by one on
    the G2 input; after each, the oracle's populations must be the reference's (direction sums,
    sums of squares and point samples of all four lattices on the y rows {0,3,5})."""
    import importlib.util

    g = _need("ref_g4.npz")
    g2 = _need("ref_g2.npz")
    spec = importlib.util.spec_from_file_location("pack_golden", os.path.join(os.path.dirname(golden_path("x")), "pack_golden.py"))
    pk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pk)
    p = _ref_grid(O)
    o = O.Oracle(p)
    o.gpu_initialization()
    o.set_fields({k: g2["input_" + k] for k in ("rho", "c", "cn", "T", "ux", "uy", "uz")})
    o.fast_poisson(float(g["shift"]))
    o.init_equilibrium()
    zs, xs = list(g["zsel"]), list(g["xsel"])
    for stage, call, which in pk.G4_STAGES:
        getattr(o, call)()
        got = pk.oracle_pops(o, which)
        s1, s2 = got.sum(axis=(2, 3, 4)), (got * got).sum(axis=(2, 3, 4))
        assert np.abs(s1 - g[stage + "_sum"]).max() <= 1e-12 * np.abs(g[stage + "_sum"]).max(), stage
        assert np.abs(s2 - g[stage + "_sumsq"]).max() <= 1e-12 * np.abs(g[stage + "_sumsq"]).max(), stage
        sample = got[:, :, zs][:, :, :, :, xs]
        ref = g[stage + "_sample"]
        scale = np.abs(ref).max(axis=(1, 2, 3, 4), keepdims=True)
        assert (np.abs(sample - ref) / scale).max() <= 1e-13, stage


# ---- second and third grid of the reference's own kernels -----------------------------------
# oracle/build_ref.sh NXxNYxNZ rebuilds the reference with NX, NY, NZ, Lx, Ly, Lz rewritten by sed
# in the TEMPORARY copy of LBM.h (lines 32-35, 40-42); make_golden.extra_grid /
# pack_golden.pack_extra turn its runs on the MI355X into tests/golden/ref_<grid>.npz.  130x6x19:
# rows of three 64-node tiles (the x+-1 pull across a tile boundary), NX > 128.  70x6x83: two
# tiles, a channel taller than 66 planes (serial Thomas z solve in the HIP path, NE = 164 = 4*41
# in the reference's FFT).
EXTRA_GRIDS = ["130x6x19", "70x6x83"]


def _extra(O, grid):
    g = _need(f"ref_{grid}.npz")
    nx, ny, nz = (int(v) for v in g["grid"])
    return g, O.default_params(nx, ny, nz)


def _check_race_aware(O, o, got, want, ys, where):
    """_check for a run of the reference's own kernels in which its read-after-write race may have
    gone either way: the z==0 thread reads node z=1's rest populations (LBM.cu:664-667) which the
    z=1 thread overwrites in the same launch (LBM.cu:1711-1714).  Only ux, uy, uz of plane z=0 see
    it.  Everything else is compared as usual; on plane 0 every node must carry the canonical
    (pre-collision) value or one of the 7 other outcomes (h0, hn0, temp0 of node z=1 each read
    before or after its collision), which the oracle also computes.  Returns the number of nodes
    that took a non-canonical outcome (20 of 210 at step 2 on 70x6x83: two 10-thread blocks)."""
    u = ("ux", "uy", "uz")
    cut = lambda d: {k: (v[1:] if k in u else v) for k, v in d.items()}  # noqa: E731
    _check(O.rel_l2(cut(got), cut(want)), where)
    alt = o.wall_velocity_alt()[:, :, ys, :]  # [8 outcomes][3][rows][nx]; outcome 0 is the canonical one
    scale = max(np.abs(want[k]).max() for k in u)
    d_pre = np.max([np.abs(got[k][0] - want[k][0]) for k in u], axis=0)
    d_any = np.min([np.max([np.abs(alt[m, i] - want[k][0]) for i, k in enumerate(u)], axis=0) for m in range(8)], axis=0)
    assert (d_any <= TOL_U * scale).all(), (where, "wall-plane velocity is no outcome of the race", d_pre.max(), d_any.max())
    return int((d_pre > TOL_U * scale).sum())


@pytest.mark.parametrize("grid", EXTRA_GRIDS)
def test_golden_extra_grid_G1_init_and_steps(O, grid):
    g, p = _extra(O, grid)
    o = O.Oracle(p)
    o.initialization_shifts(g["g1_init_shifts"])
    col = lambda d: {k: v[:, 0, 0] for k, v in d.items()}  # noqa: E731
    groups = {k: v for k, v in O.GROUPS.items() if k != "u"}
    _check(O.rel_l2(col(o.fields()), {k: g["g1_init_" + k] for k in O.FIELDS}, groups), "init")
    o.init_equilibrium()
    done = 0
    for i, mark in enumerate(int(m) for m in g["g1_marks"]):
        o.step_shifts(g["g1_step_shifts"][done:mark])
        done = mark
        _check(O.rel_l2(col(o.fields()), {k: g[f"g1_step{mark}_{k}"] for k in O.FIELDS}), f"step {mark}")
        want = float(g["g1_current"][i])
        assert abs(o.current() - want) <= 1e-11 * abs(want), (mark, o.current(), want)


@pytest.mark.parametrize("grid", EXTRA_GRIDS)
def test_golden_extra_grid_G2_perturbed_3d_run(O, grid):
    g, p = _extra(O, grid)
    ys = list(g["ysel"])
    o = O.Oracle(p)
    o.gpu_initialization()
    o.set_fields({k: g["g2_input_" + k] for k in ("rho", "c", "cn", "T", "ux", "uy", "uz")})
    sub = lambda d: {k: v[:, ys, :] for k, v in d.items()}  # noqa: E731
    o.fast_poisson(float(g["g2_shifts"][0]))
    _check(O.rel_l2(sub(o.fields()), {k: g["g2_step0_" + k] for k in O.FIELDS}, {"phi": ["phi"], "E": ["Ex", "Ey", "Ez"]}), "step 0")
    o.init_equilibrium()
    done = 0
    raced = 0
    for mark in (int(m) for m in g["g2_marks"]):
        o.step_shifts(g["g2_shifts"][1 + done : 1 + mark])
        done = mark
        raced += _check_race_aware(O, o, sub(o.fields()), {k: g[f"g2_step{mark}_{k}"] for k in O.FIELDS}, ys, f"step {mark}")
    print(f"{grid}: wall nodes where the reference's run took the other outcome of its race: {raced}")


@pytest.mark.parametrize("grid", EXTRA_GRIDS)
def test_golden_extra_grid_G4_per_kernel_and_G5_poisson(O, grid):
    import importlib.util

    g, p = _extra(O, grid)
    spec = importlib.util.spec_from_file_location("pack_golden", os.path.join(os.path.dirname(golden_path("x")), "pack_golden.py"))
    pk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pk)
    o = O.Oracle(p)
    o.gpu_initialization()
    o.set_fields({k: g["g2_input_" + k] for k in ("rho", "c", "cn", "T", "ux", "uy", "uz")})
    o.fast_poisson(float(g["g4_shift"]))
    o.init_equilibrium()
    zs, xs = list(g["g4_zsel"]), list(g["g4_xsel"])
    if p.nx > 65:
        assert 63 in xs and 64 in xs  # both sides of a 64-node tile boundary are sampled
    for stage, call, which in pk.G4_STAGES:
        getattr(o, call)()
        got = pk.oracle_pops(o, which)
        s1, s2 = got.sum(axis=(2, 3, 4)), (got * got).sum(axis=(2, 3, 4))
        assert np.abs(s1 - g[f"g4_{stage}_sum"]).max() <= 1e-12 * np.abs(g[f"g4_{stage}_sum"]).max(), stage
        assert np.abs(s2 - g[f"g4_{stage}_sumsq"]).max() <= 1e-12 * np.abs(g[f"g4_{stage}_sumsq"]).max(), stage
        sample = got[:, :, zs][:, :, :, :, xs]
        ref = g[f"g4_{stage}_sample"]
        scale = np.abs(ref).max(axis=(1, 2, 3, 4), keepdims=True)
        assert (np.abs(sample - ref) / scale).max() <= 1e-13, stage
    # G5
    ys = list(g["ysel"])
    o = O.Oracle(p)
    o.gpu_initialization()
    o.set_fields({"c": g["g5_input_c"], "cn": g["g5_input_cn"]})
    o.fast_poisson(float(g["g5_shift"]))
    got = {k: o.field(k)[:, ys, :] for k in ("phi", "Ex", "Ey", "Ez")}
    _check(O.rel_l2(got, {k: g["g5_out_" + k] for k in got}, {"phi": ["phi"], "E": ["Ex", "Ey", "Ez"]}), "poisson")


# ---- G8: asymmetric physics -------------------------------------------------------------------
# Every other golden has the two plates at the SAME zeta potential and the reference's default
# physics.  G8 was made with voltage != voltage2 and every knob off its default (written into the
# reference's __constant__/__device__ symbols at run time, ref_driver --set; LBM.h is not edited):
# a plate swap, a K/Kn or diffu/diffun mix-up, a forgotten Ext / TH / Ra / uw / exf would show here.
ASYM_GRIDS = ["50x8x51", "130x6x19"]


def _asym(O, grid):
    g = _need(f"ref_{grid}_g8.npz")
    nx, ny, nz = (int(v) for v in g["grid"])
    p = O.default_params(nx, ny, nz)
    if grid == "50x8x51":
        p.Lx, p.Ly, p.Lz = 0.5e-6, 0.08e-6, 0.5e-6
    for k, v in zip(g["param_names"], g["param_values"]):
        setattr(p, str(k), float(v))
    assert p.voltage != p.voltage2 and p.K != -p.Kn and p.diffu != p.diffun
    return g, p


@pytest.mark.parametrize("grid", ASYM_GRIDS)
def test_golden_G8_asymmetric_physics(O, grid):
    g, p = _asym(O, grid)
    o = O.Oracle(p)
    o.initialization_shifts(g["a1_init_shifts"])
    col = lambda d: {k: v[:, 0, 0] for k, v in d.items()}  # noqa: E731
    _check(O.rel_l2(col(o.fields()), {k: g["a1_init_" + k] for k in O.FIELDS}, {k: v for k, v in O.GROUPS.items() if k != "u"}), "init")
    o.init_equilibrium()
    done = 0
    for i, mark in enumerate(int(m) for m in g["a1_marks"]):
        o.step_shifts(g["a1_step_shifts"][done:mark])
        done = mark
        _check(O.rel_l2(col(o.fields()), {k: g[f"a1_step{mark}_{k}"] for k in O.FIELDS}), f"uniform step {mark}")
        want = float(g["a1_current"][i])
        assert abs(o.current() - want) <= 1e-11 * abs(want), (mark, o.current(), want)
    ys = list(g["ysel"])
    o = O.Oracle(p)
    o.gpu_initialization()
    o.set_fields({k: g["a2_input_" + k] for k in ("rho", "c", "cn", "T", "ux", "uy", "uz")})
    sub = lambda d: {k: v[:, ys, :] for k, v in d.items()}  # noqa: E731
    o.fast_poisson(float(g["a2_shifts"][0]))
    _check(O.rel_l2(sub(o.fields()), {k: g["a2_step0_" + k] for k in O.FIELDS}, {"phi": ["phi"], "E": ["Ex", "Ey", "Ez"]}), "step 0")
    o.init_equilibrium()
    done = 0
    for mark in (int(m) for m in g["a2_marks"]):
        o.step_shifts(g["a2_shifts"][1 + done : 1 + mark])
        done = mark
        _check_race_aware(O, o, sub(o.fields()), {k: g[f"a2_step{mark}_{k}"] for k in O.FIELDS}, ys, f"perturbed step {mark}")


_SANITIZED_RUN = r"""
import hashlib, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import oracle as O
h = hashlib.sha256()
for grid, nl, uw in (((16, 12, 17), 4, 0.0), ((10, 6, 9), 3, 0.0), ((12, 4, 8), 1, 0.0), ((14, 6, 11), 4, 0.02)):
    p = O.default_params(*grid)
    p.pb_iterations, p.n_lattices, p.uw = 15, nl, uw
    if nl < 4:
        p.Ra = 0.0
    if nl == 1:
        p.chargeinf, p.TH, p.exf = 0.0, 0.0, 1e9
    o = O.Oracle(p)
    o.threads = O.set_threads(2)
    o.initialization()
    o.set_fields(O.perturb_fields(p, o.fields()))
    o.fast_poisson()
    o.init_equilibrium()
    o.step(3)
    o.stream_collide_save()
    o.fast_poisson()
    vals = (o.current(), o.umax())
    f = o.fields()
    assert all(np.isfinite(v).all() for v in f.values()) and all(np.isfinite(v) for v in vals)
    for k in sorted(f):
        h.update(np.ascontiguousarray(f[k]).tobytes())
    o.close()
print("SHA", h.hexdigest())
"""


def test_oracle_under_sanitizers(tmp_path):
    """the reference has no sanitizer run (and one real race, LBM.cu:664-667 vs 1711-1714); the CPU oracle
    - the checker every parity test leans on - runs here once under AddressSanitizer + UndefinedBehaviorSanitizer (`make -C
    oracle asan`) on four small cases (4 / 3 / 1 lattices, a moving wall, a grid with an even plane count): no report, and the
    same bits as the ordinary build.  A child process: the sanitizer runtime has to be in the process before python starts."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    odir = os.path.join(root, "oracle")

    def runtime(name):
        pth = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
        return pth if os.path.isabs(pth) and os.path.exists(pth) else None

    asan, ubsan = runtime("libasan.so"), runtime("libubsan.so")
    if asan is None:
        pytest.skip("this gcc has no AddressSanitizer runtime")
    subprocess.check_call(["make", "-s", "-C", odir, "asan"])
    subprocess.check_call(["make", "-s", "-C", odir])
    script = tmp_path / "run.py"
    script.write_text(_SANITIZED_RUN)

    def run(extra_env):
        env = dict(os.environ, OMP_NUM_THREADS="2", **extra_env)
        r = subprocess.run([sys.executable, str(script), root], env=env, capture_output=True, text=True, timeout=600)
        return r.returncode, r.stdout, r.stderr

    rc, out, err = run({"EKPNP_ORACLE_LIBRARY": os.path.join(odir, "libekpnp_oracle_asan.so"),
                        "LD_PRELOAD": asan + ((" " + ubsan) if ubsan else ""),
                        "ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1:abort_on_error=0",  # (python itself leaks by design)
                        "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"})
    assert rc == 0, err[-3000:]
    assert "runtime error" not in err and "AddressSanitizer" not in err, err[-3000:]
    rc2, out2, err2 = run({})
    assert rc2 == 0, err2[-3000:]
    sha = [ln for ln in out.splitlines() if ln.startswith("SHA")]
    assert sha and sha == [ln for ln in out2.splitlines() if ln.startswith("SHA")]
