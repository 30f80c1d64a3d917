"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on
identical inputs, and against the golden vectors of the reference's own kernels.

Tolerance: BASELINE.json's north_star asks for fields within 1e-5 rel-L2 of the reference.
The HIP kernels evaluate the same formulas with a different association of a few sums
(pairwise TRT form, FMA contraction, tridiagonal z-solve instead of the odd-extension DFT),
so the expected difference is FP64 rounding amplified over the run; the tests demand
TOL = 1e-9 after up to 50 steps (4 orders tighter than the north_star), per field GROUP
(vector fields jointly).  The velocity group gets TOL_U = 1e-7: u is the
difference of O(100) populations divided by rho*CFL = 10, i.e. a 1e-5 cancellation at
u ~ 1e-4 m/s, so one rounding of a population (1e-14) is already 1e-11 of u per step.
Every measured error is also written to gpurun_out/parity_report.json."""
import json
import os

import numpy as np
import pytest

from conftest import golden_path

pytestmark = pytest.mark.gpu
TOL = 1e-9
TOL_U = 1e-7
_REPORT = []


@pytest.fixture(scope="module", autouse=True)
def _write_report():
    yield
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_report.json"), "w") as f:
            json.dump(_REPORT, f, indent=1)
    except OSError:
        pass


def _mirror(pkg, po):
    """oracle Params -> product Params (identical layout)."""
    p = pkg.Params()
    for name, _ in p._fields_:
        setattr(p, name, getattr(po, name))
    return p


def _run_pair(pkg, O, po, steps, perturb=True, init=True, start_fields=None):
    """Run oracle and HIP from the same start; returns list of (mark, got, want)."""
    orc = O.Oracle(po)
    sol = pkg.Solver(_mirror(pkg, po))
    out = []
    try:
        if start_fields is None:
            orc.initialization()
            sol.initialization()
            e0 = O.rel_l2(sol.fields(), orc.fields(), {k: v for k, v in O.GROUPS.items() if k not in ("u",)})
            out.append(("init", e0))
            start_fields = O.perturb_fields(po, orc.fields()) if perturb else orc.fields()
        orc.set_fields(start_fields)
        sol.set_fields(start_fields)
        orc.fast_poisson()
        sol.fast_Poisson()
        out.append(("poisson", O.rel_l2(sol.fields(), orc.fields(), {"phi": ["phi"], "E": ["Ex", "Ey", "Ez"]})))
        orc.init_equilibrium()
        sol.init_equilibrium()
        done = 0
        for mark in steps:
            orc.step(mark - done)
            sol.step(mark - done)
            done = mark
            out.append((mark, O.rel_l2(sol.fields(), orc.fields())))
    finally:
        sol.close()
        orc.close()
    return out


def _assert_all(res, tol=TOL, skip_groups=(), name=None):
    import inspect

    name = name or inspect.stack()[1].function
    for mark, err in res:
        _REPORT.append({"test": name, "mark": str(mark), "rel_l2": err})
        bad = {k: v for k, v in err.items() if k not in skip_groups and not (v <= (TOL_U if k == "u" else tol))}
        assert not bad, (mark, err)


def test_perturbed_small_grid_1_2_50_steps(pkg, O):
    po = O.default_params(16, 12, 17)
    po.pb_iterations = 40
    _assert_all(_run_pair(pkg, O, po, [1, 2, 50]))


def test_reference_default_grid_full_init_20_steps(pkg, O):
    """50x8x51, all 501 PB sweeps (LBM.cu:89), x-y uniform start like the reference's own run.
    u is compared on the perturbed cases; here |u| is rounding noise for the first steps."""
    po = O.default_params(50, 8, 51)
    po.Lx, po.Ly, po.Lz = 0.5e-6, 0.08e-6, 0.5e-6
    res = _run_pair(pkg, O, po, [1, 5, 20], perturb=False)
    _assert_all(res, skip_groups=("u",))
    assert res[-1][1]["u"] < 1e-6


@pytest.mark.parametrize("shape", [(80, 6, 9), (130, 4, 8), (64, 3, 6), (1, 1, 5), (7, 5, 4), (20, 4, 66), (20, 4, 67), (128, 2, 5)])
def test_ragged_and_tiny_grids(pkg, O, shape):
    """nx not a multiple of the 64-lane wave, nx > 64 with a partial last segment, nx == 1, the last
    channel height the cyclic-reduction z solve takes (66 planes) and the first one the serial
    sweep takes (67), rows of exactly two tiles."""
    po = O.default_params(*shape)
    po.pb_iterations = 10
    _assert_all(_run_pair(pkg, O, po, [1, 3]))


def test_three_lattices_no_thermal(pkg, O):
    """cfg2 physics: f + h + hn, Ra = 0 (temperature cannot feed back, LBM.cu:637)."""
    po = O.default_params(24, 10, 13)
    po.pb_iterations = 20
    po.Ra = 0.0
    po.n_lattices = 3
    res = _run_pair(pkg, O, po, [1, 10])
    _assert_all(res, skip_groups=("T",))


def test_fluid_only_poiseuille(pkg, O):
    """cfg1 physics: f only, body-force driven channel; also checks the K1 profile on the GPU."""
    po = O.default_params(8, 4, 33)
    po.exf, po.chargeinf, po.Ra, po.TH, po.pb_iterations = 1e9, 0.0, 0.0, 0.0, 2
    po.n_lattices = 1
    orc = O.Oracle(po)
    orc.initialization(); orc.init_equilibrium(); orc.step(300)
    p = _mirror(pkg, po)
    with pkg.Solver(p) as s:
        s.initialization(); s.init_equilibrium(); s.step(300)
        e = O.rel_l2(s.fields(), orc.fields(), {"rho": ["rho"], "u": ["ux", "uy", "uz"]})
        assert max(e.values()) < TOL, e
        s.step(11700)
        ux = s.get_field("ux")[:, 0, 0]
    z = np.arange(po.nz) * po.dz
    ana = po.exf / (2 * po.rho0 * po.nu) * (z - 0.5 * po.dz) * ((po.nz - 1.5) * po.dz - z)
    assert np.linalg.norm(ux[1:-1] - ana[1:-1]) / np.linalg.norm(ana[1:-1]) < 1e-3
    assert abs(ux[0] + ux[1]) < 1e-12 * abs(ux[1])


def test_cfg1_at_its_own_size_on_the_hip_path(pkg, O):
    """BASELINE.json configs[0] as it is written: 64x64x64, fluid lattice only (`k_collide_bulk<1, ...>` and the one-lattice
    plates), exf = 1e9, chargeinf = Ra = TH = 0 - the workload bench.py's cpu_baseline times on the oracle, here through
    the HIP path.  200 steps against the oracle at TOL, then the fluid wall rule cfg1 isolates
    (LBM.cu:1848-1961): rho(0) ux(0) = -rho(1) ux(1) to the last bit or two (the z == 0 override, LBM.cu:663-801), and the total mass of the lattice
    conserved to rounding (1e-13).  A second run starts from the x-y structured perturbation so that a
    swapped or mirrored direction of the one-lattice instantiation cannot hide behind the x-y uniform channel."""
    po = O.default_params(64, 64, 64)
    po.n_lattices, po.chargeinf, po.Ra, po.TH, po.exf, po.pb_iterations = 1, 0.0, 0.0, 0.0, 1e9, 1
    groups = {"rho": ["rho"], "u": ["ux", "uy", "uz"]}
    orc = O.Oracle(po)
    try:
        orc.initialization(); orc.init_equilibrium()
        with pkg.Solver(_mirror(pkg, po)) as s:
            s.initialization(); s.init_equilibrium()
            s.step(1); orc.step(1)
            m0 = float(s.get_field("rho").sum(dtype=np.float64))
            for mark, n in ((50, 49), (200, 150)):
                s.step(n); orc.step(n)
                e = O.rel_l2(s.fields(), orc.fields(), groups)
                _REPORT.append({"test": "test_cfg1_at_its_own_size_on_the_hip_path", "mark": str(mark), "rel_l2": e})
                assert e["rho"] <= TOL and e["u"] <= TOL_U, (mark, e)
            f = s.fields()
            # ux(0) = -ux(1) up to the ratio rho(1)/rho(0): the override divides node 1's momentum by node 0's density
            # (`rhoinvm = 1.0/rho`, LBM.cu:780), and the two densities differ in their last bits (oracle: 2.9e-15)
            assert np.abs(f["ux"]).max() > 0 and np.abs(f["ux"][0] + f["ux"][1]).max() <= 1e-13 * np.abs(f["ux"][1]).max()
            assert np.abs(f["ux"][0] * f["rho"][0] + f["ux"][1] * f["rho"][1]).max() <= 4e-16 * np.abs(f["ux"][1] * f["rho"][1]).max()
            assert abs(float(f["rho"].sum(dtype=np.float64)) - m0) <= 1e-12 * abs(m0)
            # the Poiseuille profile is on its way: x-y uniform, parabolic sign (fastest in the middle)
            assert np.ptp(f["ux"][32]) <= 1e-9 * abs(f["ux"][32]).max() and f["ux"][32, 0, 0] > f["ux"][2, 0, 0] > 0
        # x-y structure
        start = O.perturb_fields(po, orc.fields())
        orc.set_fields(start); orc.init_equilibrium(); orc.step(20)
        with pkg.Solver(_mirror(pkg, po)) as s:
            s.set_fields(start); s.init_equilibrium(); s.step(20)
            e = O.rel_l2(s.fields(), orc.fields(), groups)
            _REPORT.append({"test": "test_cfg1_at_its_own_size_on_the_hip_path", "mark": "perturbed 20", "rel_l2": e})
            assert e["rho"] <= TOL and e["u"] <= TOL_U, e
    finally:
        orc.close()


@pytest.mark.parametrize("shape,in_place,nl", [((48, 10, 24), 0, 4), ((48, 10, 24), 1, 4), ((130, 6, 19), 0, 3), ((256, 256, 40), 0, 4)])
def test_batch_moments_changes_no_visible_bit(pkg, O, shape, in_place, nl):
    """The opt-in knob "batch_moments": inside one ekpnp_step(n) call only the last step stores rho, u, c, cn, T (LBM.cu:807-813
    stores them every step).  Everything a caller can see after the call - all eleven fields, the diagnostics, and the run's
    continuation - must hold the same bits as without the knob: small lattices (k_collide_all, hipGraph replay), in place,
    three lattices, and a lattice large enough for the separate bulk kernel (256 x 256 x 40)."""
    po = O.default_params(*shape)
    po.pb_iterations = 6
    po.n_lattices = nl
    if nl < 4:
        po.Ra = 0.0
    p = _mirror(pkg, po)
    p.in_place = in_place
    res = []
    start = None
    for knob in (0, 1):
        with pkg.Solver(p) as s:
            s.tune("batch_moments", knob)
            s.initialization()
            if start is None:
                start = O.perturb_fields(po, s.fields())
            s.set_fields(start)
            s.fast_Poisson()
            s.init_equilibrium()
            s.step(7)
            a = {k: v.copy() for k, v in s.fields().items()}
            cur, um = s.current(), s.umax()
            s.step(1)  # a batch of one: stores
            s.step(4)
            b = {k: v.copy() for k, v in s.fields().items()}
            res.append((a, b, cur, um))
    (a0, b0, c0, u0), (a1, b1, c1, u1) = res
    for k in a0:
        assert np.array_equal(a0[k], a1[k]) and np.array_equal(b0[k], b1[k]), k
    assert c0 == c1 and u0 == u1
    if nl == 4 and shape[0] < 256:  # and the knobbed run against the oracle
        orc = O.Oracle(po)
        try:
            orc.set_fields(start); orc.fast_poisson(); orc.init_equilibrium(); orc.step(12)
            e = O.rel_l2(b1, orc.fields())
        finally:
            orc.close()
        assert all(v <= (TOL_U if k == "u" else TOL) for k, v in e.items()), e


def test_moving_wall_and_body_force(pkg, O):
    po = O.default_params(20, 6, 11)
    po.pb_iterations = 10
    po.uw, po.exf = 1e-3, 1e6
    _assert_all(_run_pair(pkg, O, po, [1, 8]))


def test_poisson_alone_random_charges(pkg, O):
    po = O.default_params(50, 8, 51)
    rng = np.random.default_rng(3)
    orc = O.Oracle(po)
    orc.gpu_initialization()
    f = orc.fields()
    f["c"] = 0.01 * (1 + 0.2 * rng.random(orc.shape))
    f["cn"] = 0.01 * (1 + 0.2 * rng.random(orc.shape))
    orc.set_fields(f)
    orc.fast_poisson()
    with pkg.Solver(_mirror(pkg, po)) as s:
        s.set_fields(f)
        s.fast_Poisson()
        e = O.rel_l2(s.fields(), orc.fields(), {"phi": ["phi"], "E": ["Ex", "Ey", "Ez"]})
    assert max(e.values()) < 1e-12, e


def test_step_api_equals_split_calls_and_time_advances(pkg, O):
    po = O.default_params(16, 8, 9)
    po.pb_iterations = 5
    p = _mirror(pkg, po)
    a, b = pkg.Solver(p), pkg.Solver(p)
    try:
        for s in (a, b):
            s.initialization()
            s.set_fields(O.perturb_fields(po, s.fields()))
            s.fast_Poisson()
            s.init_equilibrium()
        a.step(4)
        for i in range(4):
            b.stream_collide_save(i * p.dt)
            b.fast_Poisson()
        fa, fb = a.fields(), b.fields()
        for k in fa:
            assert np.array_equal(fa[k], fb[k]), k
        assert abs(a.t - 4 * p.dt) < 1e-25
    finally:
        a.close(); b.close()


def test_bind_field_uses_caller_memory(pkg, O):
    """main.cu keeps its own rho_gpu ... T_gpu (main.cu:96-106): ekpnp_bind_field makes the
    context write straight into caller-owned device arrays."""
    import torch

    po = O.default_params(16, 8, 9)
    po.pb_iterations = 3
    with pkg.Solver(_mirror(pkg, po)) as s:
        mine = torch.zeros(s.shape, dtype=torch.float64, device="cuda")
        s.bind_field("rho", mine.data_ptr())
        s.initialization(); s.init_equilibrium(); s.step(1); s.synchronize()
        assert s.field_device_ptr("rho") == mine.data_ptr()
        assert np.array_equal(mine.cpu().numpy(), s.get_field("rho"))
        assert abs(float(mine.mean()) - po.rho0) < 1e-6


def test_bound_fields_need_only_8_byte_alignment(pkg, O):
    """The reference's arrays are plain cudaMalloc'ed doubles (main.cu:96-106); a caller may as
    well hand in views into a larger buffer.  Every field bound at an address that is 8- but not
    16-byte aligned must give the same bits as the library's own arrays."""
    import torch

    po = O.default_params(18, 5, 9)
    po.pb_iterations = 3
    start = None
    with pkg.Solver(_mirror(pkg, po)) as s:
        s.initialization()
        start = O.perturb_fields(po, s.fields())
        s.set_fields(start); s.fast_Poisson(); s.init_equilibrium(); s.step(3)
        want = s.fields()
    with pkg.Solver(_mirror(pkg, po)) as s:
        n = int(np.prod(s.shape))
        stride = n + n % 2 + 2  # even, so that every view starts at an odd element
        pool = torch.zeros(len(pkg.FIELDS) * stride + 1, dtype=torch.float64, device="cuda")
        for i, k in enumerate(pkg.FIELDS):
            view = pool[i * stride + 1 : i * stride + 1 + n]
            assert view.data_ptr() % 16 == 8
            s.bind_field(k, view.data_ptr())
        s.initialization()
        s.set_fields(start); s.fast_Poisson(); s.init_equilibrium(); s.step(3)
        got = s.fields()
    for k in want:
        assert np.array_equal(got[k], want[k]), k


def test_mass_conservation_and_symmetry_at_scale(pkg, O):
    """Size-independent properties on a grid the oracle would take minutes for (256x64x66):
    total fluid mass (all nodes, walls included) is conserved, an x-y uniform start stays x-y
    uniform."""
    p = pkg.default_params(256, 64, 66)
    p.pb_iterations = 30
    with pkg.Solver(p) as s:
        s.initialization(); s.init_equilibrium()
        s.step(1)
        m0 = s.get_field("rho").sum()
        s.step(40)
        f = s.fields()
    assert all(np.isfinite(v).all() for v in f.values())
    assert abs(f["rho"].sum() / m0 - 1) < 1e-12
    for k in ("rho", "c", "cn", "phi", "T", "Ez", "uz"):
        prof = f[k][:, :1, :1]
        scale = np.abs(f[k]).max() + 1e-300
        assert np.abs(f[k] - prof).max() <= 1e-9 * scale, k


def test_cfg2_full_size_properties(pkg, O):
    """BASELINE cfg2 at its full size (256^3, f + h + hn): size-independent properties.
    (i) phi returned by fast_Poisson satisfies the discrete equation the reference solves
    (spectral in x,y: checked through its 2-D FFT; second-order FD in z, Dirichlet walls,
    poisson.cu:114-180); (ii) E is the central difference of phi bit for bit; (iii) total fluid
    mass and total ion count of the interior-to-interior exchange stay finite and mass is
    conserved; (iv) linearity of the Poisson solve in (c - cn)."""
    n = 256
    p = pkg.default_params(n, n, n)
    p.n_lattices, p.Ra = 3, 0.0
    rng = np.random.default_rng(2)
    with pkg.Solver(p) as s:
        c = 0.01 * (1 + 0.05 * rng.random(s.shape))
        cn = 0.01 * (1 + 0.05 * rng.random(s.shape))
        s.set_field("c", c); s.set_field("cn", cn)
        s.fast_Poisson()
        phi, ex, ey, ez = (s.get_field(k) for k in ("phi", "Ex", "Ey", "Ez"))
        # (ii)
        assert np.array_equal(ex, 0.5 * (np.roll(phi, 1, 2) - np.roll(phi, -1, 2)) / p.dx)
        assert np.array_equal(ey, 0.5 * (np.roll(phi, 1, 1) - np.roll(phi, -1, 1)) / p.dy)
        ezr = 0.5 * (np.roll(phi, 1, 0) - np.roll(phi, -1, 0)) / p.dz
        ezr[0], ezr[-1] = ezr[1], ezr[-2]
        assert np.array_equal(ez, ezr)
        assert np.all(phi[0] == p.voltage) and np.all(phi[-1] == p.voltage2)
        # (i) residual of  d2z phi - (kx^2 + ky^2) phi = -F (c - cn)/eps  in (kx, ky, z) space
        ph = np.fft.rfft2(phi, axes=(1, 2))
        g = np.fft.rfft2(-p.convertCtoCharge * (c - cn) / p.eps, axes=(1, 2))
        kx = 2 * np.pi * np.arange(n // 2 + 1) / p.Lx
        ky = 2 * np.pi * np.fft.fftfreq(n, d=1.0 / n) / p.Ly
        k2 = ky[:, None] ** 2 + kx[None, :] ** 2
        lhs = (ph[:-2] - 2 * ph[1:-1] + ph[2:]) / p.dz**2 - k2[None] * ph[1:-1]
        res = np.abs(lhs - g[1:-1]).max() / np.abs(g[1:-1]).max()
        assert res < 1e-9, res
        del ph, g, lhs
        # (iv)
        s.set_field("c", 2 * c - cn * 0); s.set_field("cn", 2 * cn)
        s.fast_Poisson()
        phi2 = s.get_field("phi")
        # phi = phi_walls + L(c - cn): doubling the charge doubles the deviation from the zero-charge solution
        s.set_field("c", np.zeros(s.shape)); s.set_field("cn", np.zeros(s.shape))
        s.fast_Poisson()
        phi0 = s.get_field("phi")
        assert np.abs((phi2 - phi0) - 2 * (phi - phi0)).max() < 1e-12 * np.abs(phi - phi0).max()
        del phi2, phi0
    # (iii) a short run from the Gouy-Chapman start of the bench
    import bench

    with pkg.Solver(p) as s:
        bench.gouy_chapman_state(s, p)
        bench.apply_perturbation(s, None, p)
        s.fast_Poisson(); s.init_equilibrium()
        s.step(1)
        m0 = s.get_field("rho").sum()
        s.step(20)
        f = s.fields()
        assert all(np.isfinite(v).all() for v in f.values())
        assert abs(f["rho"].sum() / m0 - 1) < 1e-12
        assert f["c"].min() > 0 and f["cn"].min() > 0


def test_cfg3_maximum_size_periodic_tiles(pkg, O):
    """BASELINE cfg3 at full size: 512^3 nodes x 4 lattices = 247 GB in two buffers per lattice (element indices of a
    population buffer beyond 2^31, byte offsets beyond 2^34; the reference's `unsigned int` index arithmetic, LBM.cu:27-30,
    ends at 165 M nodes), with x-y STRUCTURED data (the x-y uniform variant of rounds 1-3 could not see a
    swapped direction or a tile-offset error)."""
    from _periodic_tiles import periodic_tile_check

    worst = periodic_tile_check(pkg, O, 512, 0, 4, 252e9, 31)
    _REPORT.append({"test": "periodic_tiles_512x512x512", "mark": "4", "rel_max": worst})


# ---- golden vectors produced by the reference's own kernels --------------------------------
# The HIP path returns the exact (DC = 0) Poisson solution, the reference's run carries its FFT
# library's DC-mode leak (tests/test_oracle_cpu.py pins the oracle to the reference WITH the
# measured leak injected).  Directly comparable are the quantities that do not see phi.

def _need(name):
    path = golden_path(name)
    if not os.path.exists(path):
        pytest.skip(f"{name} missing")
    return np.load(path)


def _ref_grid(O):
    po = O.default_params(50, 8, 51)
    po.Lx, po.Ly, po.Lz = 0.5e-6, 0.08e-6, 0.5e-6
    return po


def test_golden_G3_hip_vs_reference_kernels_direct(pkg, O):
    g = _need("ref_g3.npz")
    po = _ref_grid(O)
    po.exf, po.chargeinf, po.Ra, po.TH = 1e9, 0.0, 0.0, 0.0
    with pkg.Solver(_mirror(pkg, po)) as s:
        s.initialization()
        s.init_equilibrium()
        done = 0
        for mark in (int(m) for m in g["marks"]):
            s.step(mark - done)
            done = mark
            f = {k: s.get_field(k)[:, [0, 3, 5], :] for k in ("rho", "ux", "uy", "uz")}
            err = O.rel_l2(f, {k: g[f"step{mark}_{k}"] for k in f}, {"rho": ["rho"], "u": ["ux", "uy", "uz"]})
            _REPORT.append({"test": "golden_G3_direct", "mark": str(mark), "rel_l2": err})
            assert err["rho"] < 1e-12 and err["u"] < 1e-8, (mark, err)


def test_golden_G7_hip_moving_wall_vs_reference_kernels_direct(pkg, O):
    g = _need("ref_g7.npz")
    po = _ref_grid(O)
    po.uw, po.chargeinf, po.Ra, po.TH = 1e-3, 0.0, 0.0, 0.0
    with pkg.Solver(_mirror(pkg, po)) as s:
        s.initialization()
        s.init_equilibrium()
        done = 0
        for mark in (int(m) for m in g["marks"]):
            s.step(mark - done)
            done = mark
            f = {k: s.get_field(k)[:, [0, 3, 5], :] for k in ("rho", "ux", "uy", "uz")}
            err = O.rel_l2(f, {k: g[f"step{mark}_{k}"] for k in f}, {"rho": ["rho"], "u": ["ux", "uy", "uz"]})
            _REPORT.append({"test": "golden_G7_direct", "mark": str(mark), "rel_l2": err})
            assert err["rho"] < 1e-12 and err["u"] < 1e-8, (mark, err)


def test_golden_G5_hip_poisson_vs_reference_modulo_dc_leak(pkg, O):
    """HIP phi == reference phi minus the reference's measured DC constant, to rounding."""
    g = _need("ref_g5.npz")
    po = _ref_grid(O)
    ys = list(g["ysel"])
    with pkg.Solver(_mirror(pkg, po)) as s:
        s.set_field("c", g["input_c"])
        s.set_field("cn", g["input_cn"])
        s.fast_Poisson()
        phi = s.get_field("phi")[:, ys, :]
    d = g["out_phi"][1:-1] - phi[1:-1]
    assert np.abs(d - float(g["shift"])).max() < 1e-15
    assert np.array_equal(phi[0], g["out_phi"][0]) and np.array_equal(phi[-1], g["out_phi"][-1])


def test_golden_G2_hip_first_step_moments_vs_reference(pkg, O):
    """c, cn, rho, T written by the first collide (LBM.cu:807-813) do not depend on phi."""
    g = _need("ref_g2.npz")
    po = _ref_grid(O)
    ys = list(g["ysel"])
    with pkg.Solver(_mirror(pkg, po)) as s:
        s.set_fields({k: g["input_" + k] for k in ("rho", "c", "cn", "T", "ux", "uy", "uz")})
        s.fast_Poisson()
        s.init_equilibrium()
        s.step(1)
        f = {k: s.get_field(k)[:, ys, :] for k in ("rho", "c", "cn", "T")}
    err = O.rel_l2(f, {k: g["step1_" + k] for k in f}, {k: [k] for k in f})
    _REPORT.append({"test": "golden_G2_step1_moments", "mark": "1", "rel_l2": err})
    assert max(err.values()) < 1e-13, err


def test_graph_replay_is_bitwise_the_eager_path(pkg, O, monkeypatch):
    """ekpnp_step replays a captured 2-step hipGraph on launch-bound lattices; the results must be
    the eager ones bit for bit, for even and odd step counts and across re-captures."""
    po = O.default_params(24, 8, 13)
    po.pb_iterations = 10
    p = _mirror(pkg, po)
    outs = []
    for mode in ("graph", "eager"):
        with pkg.Solver(p) as s:
            s.initialization()
            s.set_fields(O.perturb_fields(po, s.fields()))
            s.fast_Poisson(); s.init_equilibrium()
            if mode == "graph":
                s.step(11)
                assert s.graph_state() == 1, "the 2-step graph was not captured"
                s.step(8); s.stream_collide_save(); s.fast_Poisson(); s.step(7)
            else:
                for _ in range(11 + 8):
                    s.stream_collide_save(); s.fast_Poisson()
                s.stream_collide_save(); s.fast_Poisson()
                for _ in range(7):
                    s.stream_collide_save(); s.fast_Poisson()
            outs.append((s.fields(), s.t))
    for k in outs[0][0]:
        assert np.array_equal(outs[0][0][k], outs[1][0][k]), k
    assert abs(outs[0][1] - 26 * p.dt) < 1e-22


@pytest.mark.parametrize("shape", [(24, 6, 150), (16, 12, 17), (70, 3, 66), (7, 5, 4), (9, 3, 5)])
def test_in_place_mode_is_bitwise_the_two_buffer_mode(pkg, O, shape):
    """in_place = 1: one population buffer, every sweep writes the lattice 65 planes further down /
    up (bulk launches of 64 planes in z order).  Same kernels, same arithmetic per node: the
    results must be identical bit for bit, with about half the population memory.  (On these small
    lattices the two-buffer mode even uses another launch shape, k_collide_all: same bits.)"""
    po = O.default_params(*shape)
    po.pb_iterations = 12
    outs, mem = [], []
    for mode in (0, 1):
        p = _mirror(pkg, po)
        p.in_place = mode
        with pkg.Solver(p) as s:
            s.initialization()
            s.set_fields(O.perturb_fields(po, s.fields()))
            s.fast_Poisson(); s.init_equilibrium()
            s.step(7)
            s.stream_collide_save(); s.fast_Poisson()
            s.init_equilibrium()  # restart from the fields in the other parity
            s.step(4)
            outs.append(s.fields())
            mem.append(s.device_bytes())
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k
    assert mem[1] < mem[0]


def test_merged_wall_kernel_equals_separate_launches(pkg, O):
    """Lattices of up to 4 M nodes collide their plates and bulk in one launch (k_collide_all: one
    kernel less in the dependent chain, 0.039 -> 0.030 ms per step on the reference's 50x8x51).  Every
    node must get the same bits as from the separate k_collide_bulk / k_collide_wall launches (the
    fused multiply-adds of the velocity and force formulas are written out, so the compiler has no
    choice that could differ between the kernels)."""
    po = O.default_params(50, 8, 51)
    po.pb_iterations = 20
    outs = []
    for merged in (1, 0):
        with pkg.Solver(_mirror(pkg, po)) as s:
            s.tune("merged_walls", merged)
            s.initialization()
            s.set_fields(O.perturb_fields(po, s.fields()))
            s.fast_Poisson(); s.init_equilibrium()
            s.step(21)
            outs.append(s.fields())
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k


def test_band_order_of_the_sweep_changes_no_bit(pkg, O):
    """ekpnp_tune "bulk_yband" (round 5): the interior sweep takes bands of 128 rows of every plane, band after band, instead of
    plane after plane, so that a phi row's three uses (as z + 1, z, z - 1) lie within the Infinity Cache's reach (the default
    on planes whose sweep moves more than 192 MiB: cfg3 - cfg5).  Another order of the workgroups, the same arithmetic per
    node: every field must come out bit for bit the same - bands of 64 and 128 rows against plane order, one context
    (separate bulk launch forced: small lattices use the merged kernel, which has no bands) and two slabs; and the band
    order against the ORACLE."""
    po = O.default_params(64, 256, 14)
    po.pb_iterations = 10
    outs = {}
    start = None
    for band in (0, 64, 128):
        for nslabs in (1, 2):
            with (pkg.Solver(_mirror(pkg, po)) if nslabs == 1 else pkg.Group(_mirror(pkg, po), 2, devices=[0, 0])) as s:
                if nslabs == 1:
                    s.tune("merged_walls", 0)
                s.tune("bulk_yband", band)
                if nslabs == 1:
                    assert s.pass_order()["band_rows"] == band
                s.initialization()
                if start is None:
                    start = O.perturb_fields(po, s.fields())
                s.set_fields(start)
                s.fast_Poisson(); s.init_equilibrium()
                s.step(7)
                outs[band, nslabs] = s.fields()
    pi = _mirror(pkg, po)
    pi.in_place = 1  # in place: bands inside each of the sweep's launches of nzl / 4 planes
    for band in (0, 64):
        with pkg.Solver(pi) as s:
            s.tune("bulk_yband", band)
            s.initialization()
            s.set_fields(start)
            s.fast_Poisson(); s.init_equilibrium()
            s.step(7)
            outs[band, "in place"] = s.fields()
    for key, f in outs.items():
        for k in f:
            # (one context and two slabs solve the z system in different elimination orders: each is compared with its own plane order)
            assert np.isfinite(f[k]).all() and np.array_equal(outs[0, key[1]][k], f[k]), (key, k)
    orc = O.Oracle(po)
    try:
        orc.initialization()
        orc.set_fields(start)
        orc.fast_poisson(); orc.init_equilibrium()
        orc.step(7)
        _assert_all([(7, O.rel_l2(outs[128, 1], orc.fields()))], name="band_order_vs_oracle")
    finally:
        orc.close()


def test_in_place_mode_vs_oracle(pkg, O):
    po = O.default_params(20, 8, 140)
    po.pb_iterations = 10
    po.in_place = 1
    res = _run_pair(pkg, O, po, [1, 2, 9])
    _assert_all(res)
    p = _mirror(pkg, po)
    p.in_place = 2
    with pytest.raises(pkg.EkpnpError):
        pkg.Solver(p)


def test_fast_poisson_same_bits_from_fused_and_from_reread_rhs(pkg, O):
    """The collide writes the Poisson right-hand side from its registers; ekpnp_fast_poisson may use
    it or rebuild it from c, cn (after set_field, invalidate_rhs, or whenever c / cn are
    caller-bound).  Both producers share ONE expression (poisson.cu:121-135 order), so phi and E
    must be the same bits whichever path ran."""
    import torch

    po = O.default_params(70, 6, 19)
    po.pb_iterations = 8
    p = _mirror(pkg, po)
    res = []
    for mode in ("fused", "set_field", "invalidate", "bound"):
        with pkg.Solver(p) as s:
            keep = None
            if mode == "bound":
                keep = [torch.zeros(s.shape, dtype=torch.float64, device="cuda") for _ in range(2)]
                s.bind_field("c", keep[0].data_ptr())
                s.bind_field("cn", keep[1].data_ptr())
            s.initialization()
            s.set_fields(O.perturb_fields(po, s.fields()))
            s.fast_Poisson(); s.init_equilibrium()
            for _ in range(3):
                s.stream_collide_save()
                if mode == "set_field":
                    s.set_field("c", s.get_field("c"))
                elif mode == "invalidate":
                    s.invalidate_rhs()
                s.fast_Poisson()
            res.append(s.fields())
    for other in res[1:]:
        for k in res[0]:
            assert np.array_equal(res[0][k], other[k]), k


def test_bound_concentrations_are_read_at_call_time(pkg, O):
    """The reference's fast_Poisson reads charge_gpu / chargen_gpu when it is called
    (poisson.cu:83).  A host that changes its own (bound) c array on the device between
    stream_collide_save and fast_Poisson must get the potential of the CHANGED charge."""
    import torch

    po = O.default_params(24, 6, 17)
    po.pb_iterations = 5
    p = _mirror(pkg, po)
    with pkg.Solver(p) as s:
        c = torch.zeros(s.shape, dtype=torch.float64, device="cuda")
        cn = torch.zeros(s.shape, dtype=torch.float64, device="cuda")
        s.bind_field("c", c.data_ptr()); s.bind_field("cn", cn.data_ptr())
        s.initialization(); s.init_equilibrium()
        s.stream_collide_save(); s.synchronize()
        c.mul_(1.25)  # a custom source kernel of the host, on the device
        torch.cuda.synchronize()
        s.fast_Poisson()
        got = s.get_field("phi")
        want_c, want_cn = c.cpu().numpy(), cn.cpu().numpy()
    orc = O.Oracle(po)
    orc.gpu_initialization()
    orc.set_fields({"c": want_c, "cn": want_cn})
    orc.fast_poisson()
    assert np.abs(got - orc.field("phi")).max() <= 1e-12 * np.abs(got).max()


def test_bench_start_profile_from_the_product_vs_gouy_chapman(pkg):
    """bench.py builds its tall-channel start with the product (ekpnp_initialization_converged on a
    narrow replica); the closed-form Gouy-Chapman double layer is only the known answer here: the
    discrete PB solution agrees with it to the discretisation error of a 9.2-node Debye length."""
    import bench

    p = pkg.default_params(64, 64, 512)
    prof, note = bench.pb_profile_from_product(pkg, p)
    assert "replica" in note and prof["phi"].shape == (512,)
    z = np.arange(512, dtype=np.float64)
    lam = np.sqrt(p.eps * p.kB * p.roomT / p.electron / (2 * p.chargeinf * p.convertCtoCharge))
    vt = p.kB * p.roomT / p.electron
    wall = lambda zeta, d: 4 * vt * np.arctanh(np.tanh(zeta / (4 * vt)) * np.exp(-d / lam))  # noqa: E731
    gc = wall(p.voltage, z * p.dz) + wall(p.voltage2, (511 - z) * p.dz)
    assert np.abs(prof["phi"] - gc).max() < 0.01 * abs(p.voltage)
    assert prof["phi"][0] == p.voltage and prof["phi"][-1] == p.voltage2
    assert np.allclose(prof["c"] * prof["cn"], p.chargeinf**2, rtol=1e-9)  # Boltzmann
    assert abs(prof["phi"][256]) < 1e-7  # the double layers do not reach the mid-plane (2 zeta exp(-128/9.2) ~ 1e-8)


@pytest.mark.parametrize("shape", [(640, 64, 52), (384, 96, 58), (1152, 32, 58), (200, 100, 110), (65, 200, 170), (1000, 30, 71), (130, 129, 127),
                                   (255, 97, 87), (2050, 12, 87)])
def test_large_lattice_row_lengths_vs_oracle(pkg, O, shape):
    """>= 2 M nodes (the large-lattice launch shapes: 16-plane marching, serial z solve) with rows
    that are a multiple of 128 but not of 512 nodes (640 = 512 + 128, 384, 1152: the two-nodes-per-lane
    phi/E kernel with a partly idle last workgroup per row) and rows that are no multiple of 128 at
    all (the one-node-per-lane kernel: 200, 65, 1000, 130, 255, 2050 nodes; odd ny, odd nz)."""
    po = O.default_params(*shape)
    po.pb_iterations = 3
    _assert_all(_run_pair(pkg, O, po, [1, 2]), name=f"large_rows_{shape[0]}")


@pytest.mark.parametrize("shape,nslabs,in_place", [((128, 128, 130), 1, 0), ((128, 128, 258), 2, 0), ((128, 128, 259), 2, 0), ((200, 100, 221), 2, 0),
                                                   ((128, 128, 130), 1, 1), ((128, 128, 259), 2, 1)])
def test_large_lattice_kernels_vs_oracle(pkg, O, shape, nslabs, in_place):
    """The kernels only large lattices use - the two-nodes-per-lane phi/E kernel (rows of a multiple
    of 128 nodes, >= 2 M nodes per context) or its one-node-per-lane sibling (200-node rows), 16-plane
    marching, the serial Thomas z solve - against the oracle: one context, and two slabs (even and
    uneven) where the same kernels see the phi halo planes and the distributed z solve."""
    po = O.default_params(*shape)
    po.pb_iterations = 3
    orc = O.Oracle(po)
    orc.initialization()
    start = O.perturb_fields(po, orc.fields())
    orc.set_fields(start); orc.fast_poisson()
    pois = orc.fields()
    orc.init_equilibrium(); orc.step(3)
    want = orc.fields()
    orc.close()
    p = _mirror(pkg, po)
    p.in_place = in_place
    with (pkg.Solver(p) if nslabs == 1 else pkg.Group(p, nslabs, devices=[0] * nslabs)) as s:
        s.initialization()
        s.set_fields(start); s.fast_Poisson()
        e0 = O.rel_l2(s.fields(), pois, {"phi": ["phi"], "E": ["Ex", "Ey", "Ez"]})
        s.init_equilibrium(); s.step(3)
        e3 = O.rel_l2(s.fields(), want)
    _assert_all([("poisson", e0), (3, e3)], name=f"large_lattice_{shape[0]}x{shape[2]}_{nslabs}_{in_place}")


@pytest.mark.parametrize("case", ["cfg2", "cfg3_width"])
def test_full_size_vs_oracle(pkg, O, case):
    """Against the oracle itself at BASELINE sizes.  cfg2 at its FULL size (256x256x256, f + h + hn,
    Ra = 0: 16.8 M nodes) and cfg3's full 512x512 planes, all four lattices, on 98 planes (25.7 M nodes:
    the own row / column transforms, k_tridiag_part<8,16>, the lean bulk / plate kernels, the two-nodes-per-lane
    phi / E kernel on demand; 130 planes until round 3 - the oracle's share of this test is what the suite's time
    budget pays for; the full 512 planes would need 250 GB of host memory for the oracle, and the z extent of cfg3
    is covered by test_cfg3_maximum_size_periodic_tiles): one Poisson solve and two steps from the bench's start state."""
    import bench

    shape, nl = ((256, 256, 256), 3) if case == "cfg2" else ((512, 512, 98), 4)
    po = O.default_params(*shape)
    if nl == 3:
        po.Ra = 0.0
    p = _mirror(pkg, po)
    p.n_lattices = nl
    with pkg.Solver(p) as s:
        prof, _ = bench.pb_profile_from_product(pkg, p)
        bench.product_pb_state(s, p, prof)
        bench.apply_perturbation(s, None, p)
        start = s.fields()
        s.fast_Poisson()
        pois = {k: s.get_field(k) for k in ("phi", "Ex", "Ey", "Ez")}
        s.init_equilibrium()
        s.step(2)
        got = s.fields()
    orc = O.Oracle(po)
    orc.set_fields(start)
    del start
    orc.fast_poisson()
    e0 = O.rel_l2(pois, {k: orc.field(k) for k in pois}, {"phi": ["phi"], "E": ["Ex", "Ey", "Ez"]})
    orc.init_equilibrium()
    orc.step(2)
    e2 = O.rel_l2(got, orc.fields(), {k: v for k, v in O.GROUPS.items() if nl == 4 or k != "T"})
    orc.close()
    _assert_all([("poisson", e0), (2, e2)], name=f"{case}_full_size_vs_oracle")


@pytest.mark.parametrize("nslabs", [1, 3])
def test_asymmetric_physics_vs_oracle(pkg, O, nslabs):
    """voltage != voltage2 and every physics knob off its default (the G8 parameter set), ragged grid,
    one context and three uneven slabs: a plate swap or a K/Kn, diffu/diffun mix-up would show."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden", golden_path("make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    po = O.default_params(70, 5, 40)
    po.pb_iterations = 25
    for k, v in mg.ASYM.items():
        setattr(po, k, v)
    if nslabs == 1:
        _assert_all(_run_pair(pkg, O, po, [1, 4, 12]), name="asymmetric_physics")
        return
    orc = O.Oracle(po)
    orc.initialization()
    start = O.perturb_fields(po, orc.fields())
    orc.set_fields(start); orc.fast_poisson(); orc.init_equilibrium(); orc.step(8)
    with pkg.Group(_mirror(pkg, po), nslabs, devices=[0] * nslabs) as g:
        g.initialization()
        g.set_fields(start); g.fast_Poisson(); g.init_equilibrium(); g.step(8)
        _assert_all([(8, O.rel_l2(g.fields(), orc.fields()))], name="asymmetric_physics_slabs")
        assert abs(g.current() - orc.current()) <= 1e-8 * abs(orc.current())


def _drawn_case(O, seed):
    """One configuration drawn from a seeded generator: grid (ragged rows, nx below / across / beyond one 64-lane wave, channels
    on either side of the z-solve kernels' switch points), lattice count, population mode, slab count, and every physics knob
    within +-30 % of the G8 parameter set (signs kept: the runs stay stable), sometimes with the walls' roles exchanged."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden", golden_path("make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    rng = np.random.default_rng(seed)
    nx = int(rng.choice([rng.integers(1, 64), 64, rng.integers(65, 200), 128, rng.integers(129, 260)]))
    ny = int(rng.integers(1, 14))
    nz = int(rng.choice([rng.integers(4, 20), rng.integers(20, 66), 66, 67, rng.integers(68, 140)]))
    while nx * ny * nz > 400_000:  # the oracle finishes each case in seconds
        ny = max(1, ny // 2)
        if ny == 1:
            nx = max(1, nx // 2)
    po = O.default_params(nx, ny, nz)
    po.pb_iterations = int(rng.integers(3, 25))
    for k, v in mg.ASYM.items():
        setattr(po, k, float(v) * float(rng.uniform(0.7, 1.3)))
    if rng.random() < 0.5:
        po.voltage, po.voltage2 = po.voltage2, po.voltage
    if rng.random() < 0.3:
        po.uw = 0.0
    if rng.random() < 0.3:
        po.exf = 0.0
    nl = int(rng.choice([1, 3, 4, 4]))
    po.n_lattices = nl
    if nl < 4:
        po.Ra = 0.0
    if nl == 1:
        po.chargeinf, po.TH = 0.0, 0.0
    po.in_place = int(rng.random() < 0.4)
    max_slabs = max(1, min(4, nz // 4))
    nslabs = int(rng.integers(1, max_slabs + 1))
    steps = int(rng.integers(2, 9))
    return po, nslabs, steps


@pytest.mark.parametrize("seed", [5011, 5012, 5013, 5014, 5015, 5016, 5017, 5018, 5019, 5020, 5021, 5022])
def test_drawn_configurations_vs_oracle(pkg, O, seed):
    """Twelve configurations nobody chose by hand (seeded, so a failure names its case): whatever the draw - one context or
    up to four uneven slabs on device 0, two buffers or in place, 1 / 3 / 4 lattices, moving wall or not - the HIP path
    matches the oracle at the suite's tolerance after a few steps, and the wall current agrees."""
    po, nslabs, steps = _drawn_case(O, seed)
    nl = po.n_lattices
    skip = () if nl == 4 else (("T",) if nl == 3 else ("T", "c", "cn", "phi", "E"))
    groups = {k: v for k, v in O.GROUPS.items() if k not in skip}
    orc = O.Oracle(po)
    try:
        orc.initialization()
        start = O.perturb_fields(po, orc.fields())
        orc.set_fields(start)
        orc.fast_poisson()
        orc.init_equilibrium()
        orc.step(steps)
        want, want_cur = orc.fields(), orc.current()
    finally:
        orc.close()
    tag = f"drawn[{seed}]: {po.nx}x{po.ny}x{po.nz} nl={nl} in_place={po.in_place} slabs={nslabs} steps={steps}"
    if nslabs == 1:
        with pkg.Solver(_mirror(pkg, po)) as sol:
            sol.set_fields(start); sol.fast_Poisson(); sol.init_equilibrium(); sol.step(steps)
            got, cur = sol.fields(), sol.current()
    else:
        with pkg.Group(_mirror(pkg, po), nslabs, devices=[0] * nslabs) as g:
            g.set_fields(start); g.fast_Poisson(); g.init_equilibrium(); g.step(steps)
            got, cur = g.fields(), g.current()
    _assert_all([(steps, O.rel_l2(got, want, groups))], name=tag)
    if nl > 1:
        assert abs(cur - want_cur) <= 1e-6 * abs(want_cur) + 1e-24, tag


def test_random_call_sequences_vs_oracle(pkg, O):
    """The library's bookkeeping BETWEEN calls (lazy E validity, the fused right-hand side, graph capture, batch flags, buffer
    parity): 16 seeded random sequences of the C ABI's calls - step(n), the split pair, fast_Poisson alone, get_field, set_field
    of E / phi / c / cn / moments mid-run, init_equilibrium mid-run, ekpnp_tune, ekpnp_field_device_ptr, ekpnp_bind_field with
    device writes, ekpnp_invalidate_rhs, checkpoints into a new context of the other population mode or a group of slabs -
    mirrored call by call on the oracle (tests/diagnostics/api_fuzz.py ran 600 of them once: profiles/r05b_api_fuzz_600.json)."""
    import importlib.util
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("api_fuzz", os.path.join(root, "tests", "diagnostics", "api_fuzz.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    me = sys.modules[__name__]
    n_ops = 0
    for seed in range(2000, 2016):
        ok, ops, worst, err = fz.one_sequence(pkg, O, me, seed, None)
        n_ops += len(ops) - 1
        _REPORT.append({"test": "random_call_sequences", "mark": f"seed {seed}: " + " | ".join(ops), "rel_l2": worst})
        assert ok, (seed, ops, err)
    assert n_ops > 100


def test_anisotropic_spacings_vs_oracle(pkg, O):
    """dx != dy != dz and a box that is not NX dx long: the Poisson solve takes kx, ky from Lx, Ly
    (main.cu:119-136), its z operator from dz (poisson.cu:176) and E from dx, dy, dz
    (poisson.cu:53-55) separately; every other test has dx = dy = dz = 1e-8 and L = N d."""
    po = O.default_params(40, 12, 21)
    po.pb_iterations = 15
    po.dy, po.dz = 1.7e-8, 0.8e-8
    po.Lx, po.Ly, po.Lz = 40 * 1.0e-8 * 1.3, 12 * 1.7e-8, 20 * 0.8e-8
    _assert_all(_run_pair(pkg, O, po, [1, 6]), name="anisotropic_spacings")


@pytest.mark.parametrize("shape", [(128, 16, 130), (64, 24, 258), (128, 8, 514), (72, 10, 300), (16, 12, 68), (250, 6, 451)])
def test_partition_z_solve_equals_serial_sweeps(pkg, O, shape):
    """k_tridiag_part (one wavefront holds a whole z column: partition method + cyclic reduction across the
    lanes, spectrum read once) against the serial Thomas sweeps of k_tridiag on the same right-hand side:
    the same tridiagonal system in a different elimination order, equal to rounding.  Shapes: 4 and 8 rows
    per lane, channels that fill the 64 lanes exactly (258, 514 planes) or leave identity rows (130, 300,
    68, 451), rows of several tiles and of a partial one."""
    rng = np.random.default_rng(11)
    p = pkg.default_params(*shape)
    cc, cn = 0.01 * (1 + 0.5 * rng.random(shape[::-1])), 0.01 * (1 + 0.5 * rng.random(shape[::-1]))
    out = []
    for knob in (0, 2):
        with pkg.Solver(p) as s:
            s.tune("tri_partition", knob)
            s.set_field("c", cc)
            s.set_field("cn", cn)
            s.fast_Poisson()
            out.append({k: s.get_field(k) for k in ("phi", "Ex", "Ey", "Ez")})
    err = O.rel_l2(out[1], out[0], {"phi": ["phi"], "E": ["Ex", "Ey", "Ez"]})
    # rounding of two elimination orders; the low modes of a tall channel are the ill-conditioned ones (cond ~ NZ^2)
    assert err["phi"] < 1e-12 and err["E"] < 1e-10, err
    assert not np.array_equal(out[0]["phi"], out[1]["phi"])  # two different kernels did run


def test_partition_z_solve_random_channel_heights(pkg, O):
    """the same comparison over 10 seeded random channel heights between 67 and 514 planes (every remainder of
    the 64 x R slot grid: last lane full, partly full, empty) and random small cross-sections"""
    rng = np.random.default_rng(2026)
    for _ in range(10):
        nz = int(rng.integers(67, 515))
        nx, ny = int(rng.integers(1, 150)), int(rng.integers(1, 12))
        p = pkg.default_params(nx, ny, nz)
        shape = (nz, ny, nx)
        cc, cn = 0.01 * (1 + 0.5 * rng.random(shape)), 0.01 * (1 + 0.5 * rng.random(shape))
        out = []
        for knob in (0, 2):
            with pkg.Solver(p) as s:
                s.tune("tri_partition", knob)
                s.set_field("c", cc)
                s.set_field("cn", cn)
                s.fast_Poisson()
                out.append(s.get_field("phi"))
        err = np.sqrt(((out[1] - out[0]) ** 2).sum() / (out[0] ** 2).sum())
        assert err < 1e-12 and np.isfinite(out[1]).all(), (nx, ny, nz, err)


@pytest.mark.parametrize("shape,kernel", [((16, 8, 514), "k_tridiag_part<8>, all 64 lanes full"), ((24, 6, 300), "k_tridiag_part<8>, identity rows"),
                                          ((20, 4, 131), "k_tridiag_part<8,32>: two modes per wavefront"), ((12, 3, 131), "k_tridiag_part<4> (24 modes: no whole wide workgroup)"),
                                          ((20, 4, 100), "k_tridiag_part<8,16>: four modes per wavefront"), ((40, 8, 130), "k_tridiag_part<8,16>, all 16 lanes full")])
def test_partition_z_solve_vs_oracle(pkg, O, shape, kernel):
    """k_tridiag_part<8> - the z solve the cfg3 bench times - and <4> against the ORACLE (its 3-D DFT of the odd
    extension, poisson.cu:105-204), not against the serial HIP sweeps: tune("tri_partition", 2) takes the partition
    solve on lattices this small.  One Poisson solve from the perturbed start, then two full steps (each with its
    solve)."""
    po = O.default_params(*shape)
    po.pb_iterations = 2  # the reference's Picard damping diverges on channels this tall; two sweeps stay tame
    orc = O.Oracle(po)
    sol = pkg.Solver(_mirror(pkg, po))
    res = []
    try:
        sol.tune("tri_partition", 2)
        orc.initialization()
        sol.initialization()
        res.append(("init", O.rel_l2(sol.fields(), orc.fields(), {k: v for k, v in O.GROUPS.items() if k != "u"})))
        start = O.perturb_fields(po, orc.fields())
        orc.set_fields(start); sol.set_fields(start)
        orc.fast_poisson(); sol.fast_Poisson()
        res.append(("poisson", O.rel_l2(sol.fields(), orc.fields(), {"phi": ["phi"], "E": ["Ex", "Ey", "Ez"]})))
        orc.init_equilibrium(); sol.init_equilibrium()
        orc.step(2); sol.step(2)
        res.append((2, O.rel_l2(sol.fields(), orc.fields())))
        # and the serial sweeps from the same state give different bits: the partition kernel did run
        again = pkg.Solver(_mirror(pkg, po))
        try:
            again.tune("tri_partition", 0)
            again.set_fields(start); again.fast_Poisson()
            sol.set_fields(start); sol.fast_Poisson()
            assert not np.array_equal(again.get_field("phi"), sol.get_field("phi"))
        finally:
            again.close()
    finally:
        sol.close()
        orc.close()
    _assert_all(res, name=f"partition_z_solve_vs_oracle_{shape[0]}x{shape[2]}")


def test_sixteen_modes_per_workgroup_is_bitwise_the_default_z_solve(pkg, O):
    """The one A/B partner of DESIGN.md section 4's table that stayed in the library (ekpnp_tune "tri_wide", default off): 16
    wavefronts = 16 adjacent modes per workgroup on columns of more than 256 rows, 128 KB of LDS.  Same arithmetic per mode:
    phi must come out bit for bit the same - random charges, and all the charge in the two planes next to one plate."""
    shape = (128, 64, 400)
    p = pkg.default_params(*shape)
    rng = np.random.default_rng(29)
    cases = [(0.01 * (1 + 0.5 * rng.random(shape[::-1])), 0.01 * (1 + 0.5 * rng.random(shape[::-1])))]
    cc = np.full(shape[::-1], 0.01)
    cc[1:3] *= 1.0 + 50.0 * rng.random(cc[1:3].shape)
    cases.append((cc, np.full(shape[::-1], 0.01)))
    for cc, cn in cases:
        res = []
        for wide in (0, 1):
            with pkg.Solver(p) as s:
                s.tune("tri_partition", 2)
                s.tune("tri_wide", wide)
                s.set_field("c", cc); s.set_field("cn", cn)
                s.fast_Poisson()
                res.append(s.get_field("phi"))
        assert np.isfinite(res[0]).all() and np.array_equal(res[0], res[1])


def test_column_blocks_of_the_solve_are_bitwise_the_one_block_solve(pkg):
    """ekpnp_tune "poisson_blocks" (round 5): y forward, z solve and y inverse of one kx block back to back (the block is partly
    still in the Infinity Cache between them; three blocks are the default from 768 MiB of half spectrum on = cfg3).  Same
    kernels, the z-solve instantiation chosen on the whole spectrum, every mode solved by itself: phi must not change by a
    bit for any block count - 512-wide planes (the own column passes), 70 and 200 planes (k_tridiag_part<8,32>, <4>), random
    charges; block counts that divide the 33 column groups and ones that do not, and more blocks than groups."""
    for nz in (70, 200):
        shape = (512, 512, nz)
        p = pkg.default_params(*shape)
        p.n_lattices = 1
        p.chargeinf = 0.0
        p.Ra = 0.0
        rng = np.random.default_rng(31 + nz)
        cc, cn = 0.01 * (1 + 0.5 * rng.random(shape[::-1])), 0.01 * (1 + 0.5 * rng.random(shape[::-1]))
        with pkg.Solver(p) as s:
            s.tune("tri_partition", 2)
            s.set_field("c", cc); s.set_field("cn", cn)
            ref = None
            for nb, zc in ((1, 0), (0, 0), (2, 0), (3, 0), (5, 0), (11, 0), (33, 0), (200, 0), (1, 16), (3, 40), (1, 1000)):
                s.tune("poisson_blocks", nb)
                s.tune("poisson_zchunk", zc)  # (its measured A/B partner: rows + columns of one run of planes back to back)
                po_ = s.pass_order()  # (0 = the library decides: one block on a half spectrum of 0.15 - 0.43 GB; at most the 33 column groups)
                assert po_["poisson_blocks"] == (1 if nb == 0 else min(nb, 33)) and po_["poisson_zchunk"] == (zc if zc < nz - 2 else 0), (nb, zc, po_)
                s.fast_Poisson()
                phi = s.get_field("phi")
                if ref is None:
                    ref = phi
                    assert np.isfinite(ref).all() and np.abs(ref).max() > 1e-4
                assert np.array_equal(ref, phi), f"poisson_blocks = {nb}, poisson_zchunk = {zc} changed phi on {shape}"
        del cc, cn
    # the time loop (the collide's fused right-hand side, lazy E, hipGraph replay) through three blocks and through one
    shape = (512, 512, 70)
    p = pkg.default_params(*shape)
    p.n_lattices = 3
    p.Ra = 0.0
    p.pb_iterations = 2
    rng = np.random.default_rng(37)
    cc, cn = 0.01 * (1 + 0.05 * rng.random(shape[::-1])), 0.01 * (1 + 0.05 * rng.random(shape[::-1]))
    res = []
    for nb in (1, 3):
        with pkg.Solver(p) as s:
            s.tune("tri_partition", 2)
            s.tune("poisson_blocks", nb)
            s.initialization()
            s.set_field("c", cc); s.set_field("cn", cn)
            s.fast_Poisson()
            s.init_equilibrium()
            s.step(4)
            res.append({k: s.get_field(k) for k in ("phi", "c", "cn", "ux", "uz", "Ez")})
    for k, v in res[0].items():
        assert np.isfinite(v).all() and np.array_equal(v, res[1][k]), k


@pytest.mark.parametrize("dz", [1.0e-11, 1.0e-5])
def test_partition_z_solve_extreme_anisotropy_vs_oracle(pkg, O, dz):
    """The same extremes against the ORACLE (its 3-D DFT of the odd extension, the reference's algorithm, poisson.cu:75-204),
    not against the library's other z solve: 16 x 8 x 131 through k_tridiag_part (tri_partition = 2), dz/dx = 1e-3 and 1e3."""
    shape = (16, 8, 131)
    rng = np.random.default_rng(11)
    po = O.default_params(*shape)
    po.dz = dz
    po.Lz = (shape[2] - 1) * dz
    cc, cn = 0.01 * (1 + 0.5 * rng.random(shape[::-1])), 0.01 * (1 + 0.5 * rng.random(shape[::-1]))
    orc = O.Oracle(po)
    try:
        orc.set_fields({"c": cc, "cn": cn})
        orc.fast_poisson()
        want = {k: orc.field(k).copy() for k in ("phi", "Ex", "Ey", "Ez")}
    finally:
        orc.close()
    with pkg.Solver(_mirror(pkg, po)) as s:
        s.tune("tri_partition", 2)
        s.set_field("c", cc); s.set_field("cn", cn)
        s.fast_Poisson()
        got = {k: s.get_field(k) for k in ("phi", "Ex", "Ey", "Ez")}
    err = O.rel_l2(got, want, {"phi": ["phi"], "E": ["Ex", "Ey", "Ez"]})
    _REPORT.append({"test": "partition_z_solve_extreme_anisotropy_vs_oracle", "mark": str(dz), "rel_l2": err})
    # at dz = 1e-11 neighbouring planes differ by ~1e-6 of phi: a 1e-13 of phi is 1e-7 of Ez (see the self-comparison below)
    assert err["phi"] < TOL and err["E"] < (TOL if dz > 1e-9 else 1e-6), (dz, err)


@pytest.mark.parametrize("dz", [1.0e-11, 1.0e-5])
def test_partition_z_solve_extreme_anisotropy(pkg, O, dz):
    """recip() of k_tridiag_part (v_rcp_f64 + two Newton steps) against the IEEE divisions of the serial sweeps where the
    pivots are at their extremes: dz/dx = 1e-3 (every mode has b -> -2: the interface pivots fall to ~4e-3 through
    the six reduction levels, the system is ill-conditioned like NZ^2) and dz/dx = 1e3 (b down to -(2 + 4e6 pi^2):
    strongly diagonally dominant)."""
    rng = np.random.default_rng(5)
    for shape in ((64, 8, 300), (40, 6, 131)):
        p = pkg.default_params(*shape)
        p.dz = dz
        p.Lz = (shape[2] - 1) * dz
        cc, cn = 0.01 * (1 + 0.5 * rng.random(shape[::-1])), 0.01 * (1 + 0.5 * rng.random(shape[::-1]))
        out = []
        for knob in (0, 2):
            with pkg.Solver(p) as s:
                s.tune("tri_partition", knob)
                s.set_field("c", cc)
                s.set_field("cn", cn)
                s.fast_Poisson()
                out.append({k: s.get_field(k) for k in ("phi", "Ex", "Ey", "Ez")})
        assert all(np.isfinite(v).all() for v in out[1].values())
        err = O.rel_l2(out[1], out[0], {"phi": ["phi"], "E": ["Ex", "Ey", "Ez"]})
        # E is the central difference of the returned phi in both runs (bit for bit, K4): at dz = 1e-11 neighbouring planes
        # differ by ~1e-6 of phi, so phi's 4e-14 shows as ~1e-8 of Ez - the check of the solve is phi
        assert err["phi"] < 1e-11 and err["E"] < (1e-9 if dz > 1e-9 else 1e-6), (shape, dz, err)


def test_placement_search_changes_no_bit(pkg, O, tmp_path):
    """ekpnp_create times the real sweep on several population arenas (default up to 5; here EKPNP_PLACEMENT_TRIES=3) and keeps the fastest (capi.hip placement_search).
    Which arena is kept must not change a single bit of the results, and the probe sweeps must leave no trace (they run on
    an all-zero lattice and write NaN moments): the same run with EKPNP_PLACEMENT_TRIES=1 (no search) in another process
    gives identical fields.  512 x 128 x 80: 5.2 M nodes, above the 4 M-node threshold of the search."""
    import subprocess
    import sys

    code = r'''
import sys, json, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as G
pkg = G.load_package()
p = pkg.default_params(512, 128, 80); p.pb_iterations = 3
p.in_place = int(sys.argv[2])
with pkg.Solver(p) as s:
    rep = s.placement_report()
    s.initialization(); s.init_equilibrium(); s.step(3)
    np.savez(sys.argv[1], **s.fields())
print("REPORT", json.dumps(rep))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for in_place in ("0", "1"):
        outs = []
        for tries in ("3", "1"):
            out = tmp_path / f"f_{in_place}_{tries}.npz"
            r = subprocess.run([sys.executable, "-c", code, str(out), in_place], env=dict(os.environ, EKPNP_PLACEMENT_TRIES=tries), capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr[-2000:]
            rep = json.loads(r.stdout.split("REPORT", 1)[1])
            assert rep["tried"] == (3 if tries == "3" else 0) and (tries == "1" or all(v > 0 for v in rep["sweep_ms"])), rep
            outs.append(np.load(out))
        for k in outs[0].files:
            assert np.isfinite(outs[0][k]).all() and np.array_equal(outs[0][k], outs[1][k]), (in_place, k)


@pytest.mark.parametrize("shape", [(1024, 1024, 10), (512, 512, 12), (512, 1024, 10), (1024, 512, 10)])
def test_own_plane_transforms_equal_rocfft(pkg, O, monkeypatch, shape):
    """Planes of 512 / 1024 x 512 / 1024 nodes (cfg3, cfg4, cfg5) are transformed by the library's own row and column kernels
    (csrc/fft_plane.h: 2 + 2 kernels per solve; rocFFT needs 4 + 4 on 1024-long columns and is 0.13 ms per solve slower
    inside the cfg3 step); EKPNP_OWN_FFT=0 at creation keeps the rocFFT plans.  The two must give the same Poisson solve to
    rounding: random concentrations, one context (k_tridiag_pcr64 / serial z solve) and two slabs (the distributed solve),
    phi and E - square planes of both sizes and the two MIXED ones: the 256-point row pass with the 1024-row
    column pass and the second twiddle table at fft_tw + nx, and vice versa.  The own path against the ORACLE:
    tests/test_group_gpu.py::test_interior_rank_at_cfg5_width_vs_oracle (1024 wide), test_full_size_vs_oracle[cfg3_width]
    (512 wide, since round 4 the default there too)."""
    rng = np.random.default_rng(3)
    p = pkg.default_params(*shape)
    cc, cn = 0.01 * (1 + 0.5 * rng.random(shape[::-1])), 0.01 * (1 + 0.5 * rng.random(shape[::-1]))
    res = {}
    for own in ("1", "0"):
        monkeypatch.setenv("EKPNP_OWN_FFT", own)
        for nslabs in (1, 2):
            with (pkg.Solver(p) if nslabs == 1 else pkg.Group(p, 2, devices=[0, 0])) as s:
                s.set_field("c", cc)
                s.set_field("cn", cn)
                s.fast_Poisson()
                res[own, nslabs] = {k: s.get_field(k) for k in ("phi", "Ex", "Ey", "Ez")}
    # the own inverse row pass stores 16 bytes at a time into the phi array: a caller-bound phi that is only 8-byte aligned
    # (ekpnp_bind_field) must get the same bits
    import torch

    monkeypatch.setenv("EKPNP_OWN_FFT", "1")
    with pkg.Solver(p) as s:
        n = int(np.prod(s.shape))
        pool = torch.zeros(n + 3, dtype=torch.float64, device="cuda")
        view = pool[1 : 1 + n]
        assert view.data_ptr() % 16 == 8
        s.bind_field("phi", view.data_ptr())
        s.set_field("c", cc)
        s.set_field("cn", cn)
        s.fast_Poisson()
        assert np.array_equal(s.get_field("phi"), res["1", 1]["phi"]) and np.array_equal(s.get_field("Ez"), res["1", 1]["Ez"])
    for nslabs in (1, 2):
        a, b = res["1", nslabs], res["0", nslabs]
        assert all(np.isfinite(v).all() for v in a.values())
        err = O.rel_l2(a, b, {"phi": ["phi"], "E": ["Ex", "Ey", "Ez"]})
        assert err["phi"] < 1e-13 and err["E"] < 1e-11, (nslabs, err)
        assert not np.array_equal(a["phi"], b["phi"])  # two different transform paths did run
