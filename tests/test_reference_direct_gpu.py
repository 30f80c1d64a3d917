"""HIP path vs the outputs of the REFERENCE's own kernels, directly (no oracle in between), for the
coupled run where phi feeds back into every field.

The only thing that separates the two runs is the reference's DC-mode leak (poisson.cu:177: mode
(0,0,0) is divided by mu = 1 instead of being zeroed, so every fast_Poisson returns the exact
interior phi plus ONE constant, the FFT library's rounding residue).  The fixtures
carry that constant for every single solve of the reference's run (`*_shifts`, measured by
tests/golden/pack_golden.py from phi columns the reference driver traced).  The test drives the
product through its public C ABI exactly like main.cu:189-200 does - stream_collide_save,
fast_Poisson - and after every solve adds the reference's constant to the interior planes of the
returned phi and re-evaluates E from it on the host with the reference's own formula
(poisson.cu:45-69), writing both back with ekpnp_set_field.  The product is not changed in any way;
nothing under oracle/ computes a number here (only its rel-L2 helper is used).

Cases: the reference's default 50x8x51 run G1 (initialization() with all 501 PB sweeps, then
1/5/20/100 steps) and the perturbed 3-D run G2 (0/1/2/50 steps), and the same two runs on the two
extra compile-time grids 130x6x19 (three 64-node tiles per row) and 70x6x83 (serial z solve).
Tolerances as in test_parity_gpu.py: 1e-9 per field group, 1e-7 for the velocity group."""
import json
import os

import numpy as np
import pytest

from conftest import golden_path

pytestmark = pytest.mark.gpu
TOL, TOL_U = 1e-9, 1e-7
_REPORT = []


@pytest.fixture(scope="module", autouse=True)
def _write_report():
    yield
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_report_reference_direct.json"), "w") as f:
            json.dump(_REPORT, f, indent=1)
    except OSError:
        pass


def _need(name):
    path = golden_path(name)
    if not os.path.exists(path):
        pytest.skip(f"{name} missing")
    return np.load(path)


def reference_efield(phi, p):
    """gpu_efield + gpu_bc, poisson.cu:45-69: 0.5*(phi(-1) - phi(+1))/d, periodic in every axis,
    then Ez of a wall plane copies its interior neighbour."""
    ex = 0.5 * (np.roll(phi, 1, 2) - np.roll(phi, -1, 2)) / p.dx
    ey = 0.5 * (np.roll(phi, 1, 1) - np.roll(phi, -1, 1)) / p.dy
    ez = 0.5 * (np.roll(phi, 1, 0) - np.roll(phi, -1, 0)) / p.dz
    ez[0], ez[-1] = ez[1], ez[-2]
    return ex, ey, ez


def inject_leak(s, p, shift):
    """phi_ref = phi_exact + shift on the interior planes; E_ref = E(phi_ref)."""
    phi = s.get_field("phi")
    phi[1:-1] += shift
    ex, ey, ez = reference_efield(phi, p)
    s.set_field("phi", phi)
    s.set_field("Ex", ex)
    s.set_field("Ey", ey)
    s.set_field("Ez", ez)


def _check(O, name, mark, got, want, groups=None, wall_fraction=0.10):
    """Every group as usual - except that the velocity group is taken over the planes z >= 1 and
    plane z = 0 of u is checked node by node.  Reason: the reference's z==0 thread reads node z=1's
    rest populations (LBM.cu:664-667) while the z=1 thread overwrites them in the same launch
    (LBM.cu:1711-1714).  The product implements the canonical outcome (pre-collision values,
    SURVEY.md 8(c)); in the reference's own run on 70x6x83 two 10-thread blocks (20 of 210 sampled
    wall nodes, step 2) saw a mix - tests/test_oracle_cpu.py proves node by node that those values
    are one of the 8 possible outcomes.  Here: at most `wall_fraction` of the wall nodes may deviate (10 %
    on the fixtures; on a live 15.7 M-node run of the reference's kernels the scheduling is anybody's
    guess), and by no more than the race can explain (1e-4 of |u|; SURVEY measured <= 2.4e-6 rel-L2)."""
    groups = groups or O.GROUPS
    u = ("ux", "uy", "uz")
    cut = lambda d: {k: (v[1:] if k in u else v) for k, v in d.items()}  # noqa: E731
    err = O.rel_l2(cut(got), cut(want), groups)
    rec = {"test": name, "mark": str(mark), "rel_l2": err}
    if "u" in groups and all(k in got for k in u):
        scale = max(np.abs(want[k]).max() for k in u)
        d0 = np.max([np.abs(got[k][0] - want[k][0]) for k in u], axis=0)
        off = d0 > TOL_U * scale
        rec["wall_plane_u"] = {"nodes": int(d0.size), "raced_in_reference": int(off.sum()), "max_rel_dev": float(d0.max() / scale) if scale > 0 else 0.0}
        assert off.mean() <= wall_fraction and (scale == 0 or d0.max() <= 1e-4 * scale), (name, mark, rec)
    _REPORT.append(rec)
    bad = {k: v for k, v in err.items() if not v <= (TOL_U if k == "u" else TOL)}
    assert not bad, (name, mark, err)


class _Gold:
    """Uniform view of the default-grid fixtures (ref_g1.npz / ref_g2.npz) and of an extra grid's
    single file (ref_<grid>.npz, keys prefixed g1_ / g2_)."""

    def __init__(self, pkg, grid):
        if grid == "50x8x51":
            self.g1, self.g2 = _need("ref_g1.npz"), _need("ref_g2.npz")
            self.k1 = self.k2 = ""
            self.p = pkg.default_params(50, 8, 51)
            self.p.Lx, self.p.Ly, self.p.Lz = 0.5e-6, 0.08e-6, 0.5e-6  # literals of LBM.h:40-42
            self.ys = list(self.g2["ysel"])
            self.step_shifts1, self.shifts2 = self.g1["step_shifts"], self.g2["shifts"]
            self.init_shifts = self.g1["init_shifts"]
        else:
            g = _need(f"ref_{grid}.npz")
            self.g1 = self.g2 = g
            self.k1, self.k2 = "g1_", "g2_"
            nx, ny, nz = (int(v) for v in g["grid"])
            self.p = pkg.default_params(nx, ny, nz)  # Lx = NX dx ... as oracle/build_ref.sh wrote them
            self.ys = list(g["ysel"])
            self.step_shifts1, self.shifts2 = g["g1_step_shifts"], g["g2_shifts"]
            self.init_shifts = g["g1_init_shifts"]
        self.marks1 = [int(m) for m in self.g1[self.k1 + "marks"]]
        self.marks2 = [int(m) for m in self.g2[self.k2 + "marks"]]


GRIDS = ["50x8x51", "130x6x19", "70x6x83"]


@pytest.mark.parametrize("grid", GRIDS)
def test_default_run_G1_hip_vs_reference_direct(pkg, O, grid):
    """initialization() (LBM.cu:68-109, 501 PB sweeps, driven through the split entry points so
    that the leak of every sweep's solve can be injected) + the time loop of main.cu:189-200."""
    G = _Gold(pkg, grid)
    p = G.p
    name = f"G1_direct[{grid}]"
    col = lambda d: {k: v[:, 0, 0] for k, v in d.items()}  # noqa: E731  (x-y uniform run: z profiles)
    with pkg.Solver(p) as s:
        s.call("init_fields")       # gpu_initialization, LBM.cu:76
        s.call("pbe_begin")         # phi_old <- phi, LBM.cu:79-86
        for sh in G.init_shifts:    # LBM.cu:89-106
            s.call("pbe_concentrations")
            s.fast_Poisson()
            inject_leak(s, p, float(sh))
            s.call("pbe_relax")
        s.call("pbe_end")
        f = s.fields()
        for k in ("rho", "c", "cn", "phi", "T", "Ez"):  # the reference's fields are bit-exactly x-y uniform
            assert np.abs(f[k] - f[k][:, :1, :1]).max() <= 1e-12 * np.abs(f[k]).max(), k
        _check(O, name, "init", col(f), {k: G.g1[f"{G.k1}init_{k}"] for k in O.FIELDS}, {k: v for k, v in O.GROUPS.items() if k != "u"})
        assert not np.any(f["ux"]) and not np.any(f["uz"])
        s.init_equilibrium()
        done = 0
        for mark in G.marks1:
            for k in range(done, mark):
                s.stream_collide_save()
                s.fast_Poisson()
                inject_leak(s, p, float(G.step_shifts1[k]))
            done = mark
            _check(O, name, mark, col(s.fields()), {k: G.g1[f"{G.k1}step{mark}_{k}"] for k in O.FIELDS})


@pytest.mark.parametrize("grid", GRIDS)
def test_perturbed_run_G2_hip_vs_reference_direct(pkg, O, grid):
    G = _Gold(pkg, grid)
    p = G.p
    name = f"G2_direct[{grid}]"
    sub = lambda d: {k: v[:, G.ys, :] for k, v in d.items()}  # noqa: E731
    with pkg.Solver(p) as s:
        s.call("init_fields")
        s.set_fields({k: G.g2[f"{G.k2}input_{k}"] for k in ("rho", "c", "cn", "T", "ux", "uy", "uz")})
        s.fast_Poisson()
        inject_leak(s, p, float(G.shifts2[0]))
        _check(O, name, 0, sub(s.fields()), {k: G.g2[f"{G.k2}step0_{k}"] for k in O.FIELDS}, {"phi": ["phi"], "E": ["Ex", "Ey", "Ez"]})
        s.init_equilibrium()
        done = 0
        for mark in G.marks2:
            for k in range(done, mark):
                s.stream_collide_save()
                s.fast_Poisson()
                inject_leak(s, p, float(G.shifts2[1 + k]))
            done = mark
            _check(O, name, mark, sub(s.fields()), {k: G.g2[f"{G.k2}step{mark}_{k}"] for k in O.FIELDS})


@pytest.mark.parametrize("grid", GRIDS[1:])
def test_extra_grid_G5_poisson_and_first_step_moments(pkg, O, grid):
    """fast_Poisson alone on random charges (HIP phi == reference phi minus its DC constant) and the
    moments written by the first collide on the G2 input, which do not see phi at all."""
    G = _Gold(pkg, grid)
    g, p = G.g2, G.p
    with pkg.Solver(p) as s:
        s.set_field("c", g["g5_input_c"])
        s.set_field("cn", g["g5_input_cn"])
        s.fast_Poisson()
        phi = s.get_field("phi")[:, G.ys, :]
    d = g["g5_out_phi"][1:-1] - phi[1:-1]
    assert np.abs(d - float(g["g5_shift"])).max() < 1e-15
    assert np.array_equal(phi[0], g["g5_out_phi"][0]) and np.array_equal(phi[-1], g["g5_out_phi"][-1])
    with pkg.Solver(p) as s:
        s.call("init_fields")
        s.set_fields({k: g["g2_input_" + k] for k in ("rho", "c", "cn", "T", "ux", "uy", "uz")})
        s.fast_Poisson()
        s.init_equilibrium()
        s.step(1)
        f = {k: s.get_field(k)[:, G.ys, :] for k in ("rho", "c", "cn", "T")}
    err = O.rel_l2(f, {k: g["g2_step1_" + k] for k in f}, {k: [k] for k in f})
    _REPORT.append({"test": f"G2_step1_moments[{grid}]", "mark": "1", "rel_l2": err})
    assert max(err.values()) < 1e-13, err


@pytest.mark.parametrize("grid", ["50x8x51", "130x6x19"])
def test_asymmetric_physics_G8_hip_vs_reference_direct(pkg, O, grid):
    """G8: the two plates at different zeta potentials and every physics knob off its default (the
    reference's symbols written at run time).  Both runs - the x-y uniform one from initialization()
    and the perturbed 3-D one - through the public C ABI with the reference's DC constants injected."""
    g = _need(f"ref_{grid}_g8.npz")
    nx, ny, nz = (int(v) for v in g["grid"])
    p = pkg.default_params(nx, ny, nz)
    if grid == "50x8x51":
        p.Lx, p.Ly, p.Lz = 0.5e-6, 0.08e-6, 0.5e-6
    for k, v in zip(g["param_names"], g["param_values"]):
        setattr(p, str(k), float(v))
    assert p.voltage != p.voltage2
    name = f"G8_direct[{grid}]"
    col = lambda d: {k: v[:, 0, 0] for k, v in d.items()}  # noqa: E731
    with pkg.Solver(p) as s:
        s.call("init_fields")
        s.call("pbe_begin")
        for sh in g["a1_init_shifts"]:
            s.call("pbe_concentrations")
            s.fast_Poisson()
            inject_leak(s, p, float(sh))
            s.call("pbe_relax")
        s.call("pbe_end")
        _check(O, name, "init", col(s.fields()), {k: g["a1_init_" + k] for k in O.FIELDS}, {k: v for k, v in O.GROUPS.items() if k != "u"})
        s.init_equilibrium()
        done = 0
        for mark in (int(m) for m in g["a1_marks"]):
            for k in range(done, mark):
                s.stream_collide_save()
                s.fast_Poisson()
                inject_leak(s, p, float(g["a1_step_shifts"][k]))
            done = mark
            _check(O, name, f"uniform {mark}", col(s.fields()), {k: g[f"a1_step{mark}_{k}"] for k in O.FIELDS})
    ys = list(g["ysel"])
    sub = lambda d: {k: v[:, ys, :] for k, v in d.items()}  # noqa: E731
    with pkg.Solver(p) as s:
        s.call("init_fields")
        s.set_fields({k: g["a2_input_" + k] for k in ("rho", "c", "cn", "T", "ux", "uy", "uz")})
        s.fast_Poisson()
        inject_leak(s, p, float(g["a2_shifts"][0]))
        _check(O, name, "perturbed 0", sub(s.fields()), {k: g["a2_step0_" + k] for k in O.FIELDS}, {"phi": ["phi"], "E": ["Ex", "Ey", "Ez"]})
        s.init_equilibrium()
        done = 0
        for mark in (int(m) for m in g["a2_marks"]):
            for k in range(done, mark):
                s.stream_collide_save()
                s.fast_Poisson()
                inject_leak(s, p, float(g["a2_shifts"][1 + k]))
            done = mark
            _check(O, name, f"perturbed {mark}", sub(s.fields()), {k: g[f"a2_step{mark}_{k}"] for k in O.FIELDS})


@pytest.mark.parametrize("nslabs", [1, 3])
def test_reference_kernels_live_at_cfg2_scale(pkg, O, tmp_path, nslabs):
    """(nslabs = 3, round 5: the same lattice as THREE uneven z slabs of the multi-GPU path - ekpnp_group_* on device 0, halo ring,
    distributed z solve - against the reference's single-device kernels.)
    The reference's own kernels RUN HERE, on a 15.7 M-node lattice (250x250x251 - cfg2's scale; NX must be
    a multiple of the reference's 10-thread blocks), against the HIP path on the same input: one Poisson
    solve and two full steps, all 11 fields.  Uses oracle/_ref/ref_driver_250x250x251 (the reference's
    LBM.cu / poisson.cu built for gfx950 by `oracle/build_ref.sh 250x250x251`; it travels to the GPU box
    like the library's own .so) - skipped when that binary is not there.  The reference's DC constant of
    every solve is read off its own output (it must be ONE constant over the interior) and injected into
    the HIP run through the public API, exactly as in the fixture-based tests above."""
    import subprocess

    import bench

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drv = os.path.join(root, "oracle", "_ref", "ref_driver_250x250x251")
    if not os.path.exists(drv):
        pytest.skip("oracle/_ref/ref_driver_250x250x251 not built (oracle/build_ref.sh 250x250x251)")
    nx, ny, nz = 250, 250, 251
    n = nx * ny * nz
    p = pkg.default_params(nx, ny, nz)
    name = "reference_live_250x250x251" + ("" if nslabs == 1 else f"_{nslabs}_slabs")

    def read_bin(path):
        a = np.fromfile(path, dtype=np.float64)
        assert a.size == 11 * n, (path, a.size)
        return {k: a[i * n:(i + 1) * n].reshape(nz, ny, nx) for i, k in enumerate(O.FIELDS)}

    with (pkg.Solver(p) if nslabs == 1 else pkg.Group(p, nslabs, devices=[0] * nslabs)) as s:
        s.z0 = 0  # (bench's start-state helpers address a rank's planes: a whole lattice starts at plane 0)
        prof, _ = bench.pb_profile_from_product(pkg, p)
        bench.product_pb_state(s, p, prof)
        bench.apply_perturbation(s, None, p)
        start = s.fields()
        inp = tmp_path / "in.bin"
        with open(inp, "wb") as f:
            for k in O.FIELDS:
                f.write(np.ascontiguousarray(start[k]).tobytes())
        r = subprocess.run([drv, str(tmp_path), "fields", str(inp), "L", "1", "2"], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]
        os.remove(inp)

        def leak_of(ref_phi, hip_phi):
            d = ref_phi[1:-1] - hip_phi[1:-1]
            sh = float(d.mean())
            assert np.abs(d - sh).max() <= 1e-15, ("the reference's phi minus the exact solve is not one constant", np.abs(d - sh).max())
            return sh

        s.fast_Poisson()
        ref = read_bin(tmp_path / "L_step0.bin")
        inject_leak(s, p, leak_of(ref["phi"], s.get_field("phi")))
        _check(O, name, 0, {k: s.get_field(k) for k in ("phi", "Ex", "Ey", "Ez")}, ref, {"phi": ["phi"], "E": ["Ex", "Ey", "Ez"]})
        s.init_equilibrium()
        for step in (1, 2):
            s.stream_collide_save()
            s.fast_Poisson()
            ref = read_bin(tmp_path / f"L_step{step}.bin")
            inject_leak(s, p, leak_of(ref["phi"], s.get_field("phi")))
            _check(O, name, step, s.fields(), ref, wall_fraction=1.0)
            os.remove(tmp_path / f"L_step{step}.bin")


@pytest.mark.parametrize("seed", [31, 32, 33, 34, 35, 36])
def test_reference_kernels_live_drawn_physics(pkg, O, tmp_path, seed):
    """The reference's own kernels RUN HERE on physics nobody chose by hand: every symbol ref_driver can set at run time
    (hipMemcpyToSymbol, no source edit: voltage, voltage2, Ext, TH, Ra, K, Kn, diffu, diffun, nu, D, exf, uw, VC, VCn, V, VT,
    chargeinf, eps) drawn within 30 % of the G8 set, the plates' potentials exchanged now and then, moving wall / body force
    on or off - on the reference's default 50x8x51 grid (oracle/_ref/ref_driver), from a perturbed 3-D start, six steps with a
    dump after every one.  The HIP path gets the same fields and parameters through the C ABI and is compared with the
    reference's output DIRECTLY after every step (the reference's DC constant of each solve read off its own phi and injected
    as in the live test above).  Skipped when the binary is not there (build() makes it where /root/reference is)."""
    import importlib.util
    import subprocess

    import bench

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drv = os.path.join(root, "oracle", "_ref", "ref_driver")
    if not os.path.exists(drv):
        pytest.skip("oracle/_ref/ref_driver not built (oracle/build_ref.sh)")
    spec = importlib.util.spec_from_file_location("make_golden", golden_path("make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    rng = np.random.default_rng(seed)
    nx, ny, nz = 50, 8, 51
    n = nx * ny * nz
    p = pkg.default_params(nx, ny, nz)
    p.Lx, p.Ly, p.Lz = 0.5e-6, 0.08e-6, 0.5e-6  # literals of LBM.h:40-42
    p.pb_iterations = 60
    phys = {k: float(v) * float(rng.uniform(0.7, 1.3)) for k, v in mg.ASYM.items()}
    if rng.random() < 0.5:
        phys["voltage"], phys["voltage2"] = phys["voltage2"], phys["voltage"]
    if rng.random() < 0.35:
        phys["uw"] = 0.0
    if rng.random() < 0.35:
        phys["exf"] = 0.0
    for k, v in phys.items():
        setattr(p, k, v)
    name = f"reference_live_drawn_physics[{seed}]"
    marks = [1, 2, 3, 4, 5, 6]

    def read_bin(path):
        a = np.fromfile(path, dtype=np.float64)
        assert a.size == 11 * n, (path, a.size)
        return {k: a[i * n:(i + 1) * n].reshape(nz, ny, nx) for i, k in enumerate(O.FIELDS)}

    with pkg.Solver(p) as s:
        s.initialization()
        bench.apply_perturbation(s, None, p)
        start = s.fields()
        inp = tmp_path / "in.bin"
        with open(inp, "wb") as f:
            for k in O.FIELDS:
                f.write(np.ascontiguousarray(start[k]).tobytes())
        sets = []
        for k, v in phys.items():
            sets += ["--set", f"{k}={v!r}"]
        r = subprocess.run([drv, str(tmp_path), *sets, "fields", str(inp), "D", *[str(m) for m in marks]], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]

        def leak_of(ref_phi, hip_phi):
            d = ref_phi[1:-1] - hip_phi[1:-1]
            sh = float(d.mean())
            assert np.abs(d - sh).max() <= 1e-15, ("the reference's phi minus the exact solve is not one constant", np.abs(d - sh).max())
            return sh

        s.fast_Poisson()
        ref = read_bin(tmp_path / "D_step0.bin")
        inject_leak(s, p, leak_of(ref["phi"], s.get_field("phi")))
        _check(O, name, 0, {k: s.get_field(k) for k in ("phi", "Ex", "Ey", "Ez")}, ref, {"phi": ["phi"], "E": ["Ex", "Ey", "Ez"]})
        s.init_equilibrium()
        for step in marks:
            s.stream_collide_save()
            s.fast_Poisson()
            ref = read_bin(tmp_path / f"D_step{step}.bin")
            inject_leak(s, p, leak_of(ref["phi"], s.get_field("phi")))
            _check(O, name, step, s.fields(), ref, wall_fraction=1.0)


# ---- G4: the reference's POPULATIONS, kernel by kernel, against the HIP path's own state ------------------
# D3Q27 directions in the reference's numbering (LBM.cu:1983-2008): c[d] = (cx, cy, cz)
_C27 = [(0, 0, 0), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1), (1, 1, 0), (-1, -1, 0), (1, 0, 1), (-1, 0, -1),
        (0, 1, 1), (0, -1, -1), (1, -1, 0), (-1, 1, 0), (1, 0, -1), (-1, 0, 1), (0, 1, -1), (0, -1, 1), (1, 1, 1), (-1, -1, -1),
        (1, 1, -1), (-1, -1, 1), (1, -1, 1), (-1, 1, -1), (-1, 1, 1), (1, -1, -1)]


def _checkpoint_populations(path, p):
    """[lattice][z][y][x][27]: the post-collision populations of a whole-lattice EKPNPCK2 file (layout documented
    in include/ekpnp.h / io.hip: 64-byte header, 11 fields, then per lattice NZ planes of [y][x/64][27 slots][64])."""
    nx, ny, nz, nl = p.nx, p.ny, p.nz, p.n_lattices
    tiles = (nx + 63) // 64
    raw = np.fromfile(path, dtype=np.float64, offset=64)
    nf = 11 * nx * ny * nz
    pops = raw[nf:].reshape(nl, nz, ny, tiles, 27, 64)
    assert raw.size == nf + pops.size
    # since round 5 ("EKPNPCK2") direction d lies in slot 9 (c_z + 1) + 3 (c_y + 1) + (c_x + 1) of its tile
    slot = [9 * (cz + 1) + 3 * (cy + 1) + (cx + 1) for cx, cy, cz in _C27]
    assert sorted(slot) == list(range(27))
    pops = pops[:, :, :, :, slot, :]
    return np.moveaxis(pops, 4, 5).reshape(nl, nz, ny, tiles * 64, 27)[:, :, :, :nx, :]


@pytest.mark.parametrize("grid", GRIDS)
def test_G4_populations_hip_vs_reference_direct(pkg, O, grid, tmp_path):
    """SURVEY.md 8(c) G4 - populations of all four lattices dumped by the reference after each of its launches
    (LBM.cu:474-477) - against the HIP path's state, read through the public ekpnp_save_checkpoint.  The HIP path
    keeps POST-COLLISION populations and streams them by pulling in the next sweep, so on the interior planes
      * its state after a sweep IS the reference's X0 / X2 after gpu_collide_save (stage "collide", and the rest
        population X0 of stage "step1");
      * that state shifted by c_d (gpu_stream, LBM.cu:1963-2093, periodic in x and y) IS the reference's X1 after the
        whole stream_collide_save on the planes 2..NZ-3, whose sources are interior nodes (stages "step1", "bc_charge").
    The wall planes, where the reference's gpu_boundary / gpu_bc_charge rewrite populations that the HIP wall kernel
    folds into its own private form, are covered by the field-level comparisons above.  Sampled nodes: the z, y, x
    selections the fixture holds (both sides of a 64-node tile boundary on the wide grids)."""
    G = _Gold(pkg, grid)
    p = G.p
    if grid == "50x8x51":
        g4 = _need("ref_g4.npz")
        key = lambda stage: g4[stage + "_sample"]  # noqa: E731
        zs, xs, ys, shift = list(g4["zsel"]), list(g4["xsel"]), list(g4["ysel"]), float(g4["shift"])
    else:
        g4 = G.g2
        key = lambda stage: g4[f"g4_{stage}_sample"]  # noqa: E731
        zs, xs, ys, shift = list(g4["g4_zsel"]), list(g4["g4_xsel"]), list(g4["ysel"]), float(g4["g4_shift"])
    nz = p.nz
    ck = str(tmp_path / "state.ck")

    def state(s):
        s.save_checkpoint(ck)
        return _checkpoint_populations(ck, p)

    def streamed(P):  # X1[d](x) = X2[d](x - c_d)
        out = np.empty_like(P)
        for d, (cx, cy, cz) in enumerate(_C27):
            out[..., d] = np.roll(P[..., d], (cz, cy, cx), axis=(1, 2, 3))
        return out

    def compare(P, stage, zmin, zmax, dirs):
        ref = key(stage)  # [4][27][len(zs)][len(ys)][len(xs)]
        worst = 0.0
        for iz, z in enumerate(zs):
            if not zmin <= z <= zmax:
                continue
            got = P[:, z][:, ys][:, :, xs]  # [4][ys][xs][27]
            for d in dirs:
                r = ref[:, d, iz]
                scale = np.abs(ref[:, d]).max(axis=(1, 2, 3))[:, None, None]
                worst = max(worst, float((np.abs(got[..., d] - r) / scale).max()))
        return worst

    with pkg.Solver(p) as s:
        s.call("init_fields")
        s.set_fields({k: G.g2[f"{G.k2}input_{k}"] for k in ("rho", "c", "cn", "T", "ux", "uy", "uz")})
        s.fast_Poisson()
        inject_leak(s, p, shift)
        s.init_equilibrium()
        s.stream_collide_save()
        P1 = state(s)
        s.stream_collide_save()  # the reference's dump launches the four kernels again without a solve in between
        P2 = state(s)
    rec = {
        "step1_rest": compare(P1, "step1", 1, nz - 2, [0]),
        "step1_streamed": compare(streamed(P1), "step1", 2, nz - 3, range(1, 27)),
        "collide": compare(P2, "collide", 1, nz - 2, range(27)),
        "stream": compare(streamed(P2), "stream", 2, nz - 3, range(1, 27)),
        "bc_charge": compare(streamed(P2), "bc_charge", 2, nz - 3, range(1, 27)),
    }
    _REPORT.append({"test": f"G4_direct[{grid}]", "mark": "populations", "max_rel_dev": rec})
    assert sum(1 for z in zs if 2 <= z <= nz - 3) >= 1 and max(rec.values()) <= 1e-12, rec
