"""Lazy E (round 4): inside the time loop a solve leaves phi's interior planes in the phi array and does not run
k_phi_efield; the next collide forms E = 0.5*(phi(-1) - phi(+1))/d itself (gpu_efield / gpu_bc, poisson.cu:40-69, in the
very expression k_phi_efield uses), and the Ex / Ey / Ez arrays and phi's pinned plates are written only when somebody
looks.  Everything here is a bit-for-bit comparison with the eager path of rounds 1-3 (ekpnp_tune "lazy_efield" 0 /
EKPNP_LAZY_E=0), which the other test files compare with the oracle and with the reference's own kernels."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _start(pkg, O, s, p):
    s.initialization()
    s.set_fields(O.perturb_fields(p, s.fields()))
    s.fast_Poisson()
    s.init_equilibrium()


def _same(a, b, what=""):
    for k in a:
        assert np.array_equal(a[k], b[k]), (what, k, float(np.abs(a[k] - b[k]).max()))


@pytest.mark.parametrize("shape,nl,in_place", [((16, 12, 17), 4, 0), ((50, 8, 51), 4, 0), ((130, 6, 19), 4, 0), ((70, 6, 83), 3, 0),
                                               ((256, 128, 130), 4, 0), ((24, 6, 150), 4, 1), ((7, 5, 4), 4, 0), ((256, 128, 5), 4, 0)])
def test_lazy_efield_is_bitwise_the_eager_path(pkg, O, shape, nl, in_place):
    """Launch-bound lattices (one merged launch, hipGraph replay), ragged rows, three lattices (Ez formed by the fluid
    wave), a lattice large enough for the lean bulk / wall kernel pair and the two-nodes-per-lane phi / E kernel, in-place
    populations, the smallest channel (4 planes: both plates' Ez come from the two interior planes)."""
    p = pkg.default_params(*shape)
    p.pb_iterations = 8
    p.n_lattices = nl
    p.in_place = in_place
    if nl < 4:
        p.Ra = 0.0
    outs = []
    for lazy in (1, 0):
        with pkg.Solver(p) as s:
            s.tune("lazy_efield", lazy)
            _start(pkg, O, s, p)
            s.step(5)
            mid = s.get_field("Ez")  # looking materialises; the run goes on from phi as before
            s.step(4)
            s.stream_collide_save(); s.fast_Poisson()
            outs.append((s.fields(), mid))
    _same(outs[0][0], outs[1][0], shape)
    assert np.array_equal(outs[0][1], outs[1][1])


def test_lazy_efield_at_cfg2_size(pkg, O):
    """256^3, f + h + hn (BASELINE cfg2): ten steps lazily == ten steps eagerly, all 11 fields; and the E that comes out of
    the lazy path is the central difference of the phi that comes out, bit for bit (poisson.cu:45-69)."""
    p = pkg.default_params(256, 256, 256)
    p.pb_iterations = 3
    p.n_lattices = 3
    p.Ra = 0.0
    outs = []
    for lazy in (1, 0):
        with pkg.Solver(p) as s:
            s.tune("lazy_efield", lazy)
            _start(pkg, O, s, p)
            s.step(10)
            outs.append({k: s.get_field(k) for k in ("rho", "ux", "uz", "c", "cn", "phi", "Ex", "Ey", "Ez")})
    _same(outs[0], outs[1], "cfg2")
    f = outs[0]
    phi = f["phi"]
    assert np.all(phi[0] == p.voltage) and np.all(phi[-1] == p.voltage2)
    ex = 0.5 * (np.roll(phi, 1, axis=2) - np.roll(phi, -1, axis=2)) / p.dx
    ey = 0.5 * (np.roll(phi, 1, axis=1) - np.roll(phi, -1, axis=1)) / p.dy
    ez = np.empty_like(phi)
    ez[1:-1] = 0.5 * (phi[:-2] - phi[2:]) / p.dz
    ez[0], ez[-1] = ez[1], ez[-2]
    assert np.array_equal(f["Ex"], ex) and np.array_equal(f["Ey"], ey) and np.array_equal(f["Ez"], ez)


def test_fields_set_from_outside_are_honoured(pkg, O):
    """The reference's collide reads Ex / Ey / Ez, whatever wrote them (LBM.cu:632-637).  A caller that overwrites E (or phi:
    no effect on the collide) between fast_Poisson and stream_collide_save gets the same bits lazily and eagerly, and so
    does one that writes E on the device through ekpnp_field_device_ptr - which switches the context to the eager path."""
    import torch

    p = pkg.default_params(40, 6, 21)
    p.pb_iterations = 5
    rng = np.random.default_rng(3)
    ex_new = 1.0e3 * rng.standard_normal((21, 6, 40))
    phi_new = -5.0e-3 * rng.random((21, 6, 40))
    outs = []
    for lazy in (1, 0):
        with pkg.Solver(p) as s:
            s.tune("lazy_efield", lazy)
            _start(pkg, O, s, p)
            s.step(3)
            s.set_field("Ex", ex_new)           # E from outside: Ey, Ez must be the last solve's
            s.step(2)
            s.set_field("phi", phi_new)         # phi from outside: the collide must still see the last solve's E
            s.stream_collide_save()
            a = s.fields()
            s.fast_Poisson()
            s.step(2)
            ptr = s.field_device_ptr("Ez")      # exposure: from here on the arrays are written by every solve
            ez = torch.as_tensor(_Wrap(ptr, (21, 6, 40)), device="cuda")  # zero copy: a view of the library's own Ez array
            s.synchronize()
            ez.mul_(1.5)
            torch.cuda.synchronize()
            s.stream_collide_save(); s.fast_Poisson(); s.step(2)
            outs.append((a, s.fields()))
    _same(outs[0][0], outs[1][0], "after set_field")
    _same(outs[0][1], outs[1][1], "after a device-side edit")


class _Wrap:
    """a device pointer for torch.as_tensor: __cuda_array_interface__ (FP64, C order)"""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def test_bound_e_arrays_are_written_by_every_solve(pkg, O):
    """main.cu owns ex_gpu / ey_gpu / ez_gpu (main.cu:103-105): once one of phi / E is caller-bound the context is eager -
    the caller's array holds the new E right after the step, without any call that 'looks'."""
    import torch

    p = pkg.default_params(32, 8, 17)
    p.pb_iterations = 5
    with pkg.Solver(p) as ref:
        ref.tune("lazy_efield", 0)
        _start(pkg, O, ref, p)
        ref.step(4)
        want = ref.fields()
    with pkg.Solver(p) as s:
        mine = torch.zeros(s.shape, dtype=torch.float64, device="cuda")
        s.bind_field("Ez", mine.data_ptr())
        _start(pkg, O, s, p)
        s.step(4)
        torch.cuda.synchronize()
        import ctypes  # wait for the context's stream without calling anything that could materialise
        hip = ctypes.CDLL("libamdhip64.so")
        assert hip.hipDeviceSynchronize() == 0
        assert np.array_equal(mine.cpu().numpy(), want["Ez"])
        _same(s.fields(), want, "bound Ez")


@pytest.mark.parametrize("nslabs,in_place", [(3, 0), (2, 1)])
def test_lazy_efield_on_slabs(pkg, O, monkeypatch, nslabs, in_place):
    """z slabs: stage 3 of the distributed solve ends after the phi planes have been exchanged; the boundary planes of the
    next collide take Ez from the neighbour's plane.  Lazy group == eager group == (to rounding) one context."""
    p = pkg.default_params(70, 5, 45)
    p.pb_iterations = 6
    p.in_place = in_place
    outs = []
    for lazy in ("1", "0"):
        monkeypatch.setenv("EKPNP_LAZY_E", lazy)
        with pkg.Group(p, nslabs) as g:
            g.initialization()
            g.set_fields(O.perturb_fields(p, g.fields()))
            g.fast_Poisson(); g.init_equilibrium()
            g.step(4)
            cur = g.current()
            g.step(3)
            outs.append((g.fields(), cur))
    _same(outs[0][0], outs[1][0], "group")
    assert outs[0][1] == outs[1][1]
    monkeypatch.delenv("EKPNP_LAZY_E")
    with pkg.Solver(p) as s:
        _start(pkg, O, s, p)
        s.step(7)
        one = s.fields()
    err = O.rel_l2(outs[0][0], one)
    assert all(v < (1e-7 if k == "u" else 1e-9) for k, v in err.items()), err


def test_writers_and_checkpoint_see_current_fields(pkg, O, tmp_path):
    """save_data_end, the lossless state file and the full checkpoint written right after a lazy step hold the same bytes
    as after an eager one, and a run continued from the checkpoint stays bit-identical."""
    p = pkg.default_params(20, 6, 13)
    p.pb_iterations = 5
    files = []
    for lazy in (1, 0):
        with pkg.Solver(p) as s:
            s.tune("lazy_efield", lazy)
            _start(pkg, O, s, p)
            s.step(6)
            d = tmp_path / f"lazy{lazy}"
            d.mkdir()
            s.save_data_end(str(d / "data_end.dat"), 1.0)
            s.save_state(str(d / "state.bin"), 1.0)
            s.save_checkpoint(str(d / "ck.bin"))
            s.step(3)
            after = s.fields()
        with pkg.Solver(p) as s2:
            s2.tune("lazy_efield", lazy)
            s2.load_checkpoint(str(d / "ck.bin"))
            s2.step(3)
            _same(s2.fields(), after, "continued from the checkpoint")
        files.append({n: (d / n).read_bytes() for n in ("data_end.dat", "state.bin", "ck.bin")})
    for n in files[0]:
        assert files[0][n] == files[1][n], n
