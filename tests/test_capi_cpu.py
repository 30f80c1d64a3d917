"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/ekpnp.h declares, mirrors the reference's defaults, and fails LOUDLY without a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    return os.path.exists("/dev/kfd")


def test_header_symbols_are_exported(pkg):
    lib = pkg.load_library()
    names = pkg.exported_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"libekpnp.so does not export {n}"


def test_header_cites_reference_lines():
    txt = open(os.path.join(ROOT, "include", "ekpnp.h")).read()
    for ref in ("LBM.h:159", "LBM.h:162-163", "LBM.h:165-166", "LBM.h:176", "main.cu:189-200"):
        assert ref in txt


def test_no_torch_types_in_abi():
    txt = open(os.path.join(ROOT, "include", "ekpnp.h")).read()
    code = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)  # declarations only, comments stripped
    assert "torch" not in code.lower() and "at::" not in code and "std::" not in code


def test_default_params_match_reference_header(pkg, O):
    p = pkg.default_params(50, 8, 51)
    q = O.default_params(50, 8, 51)
    for name, _ in p._fields_:
        assert getattr(p, name) == getattr(q, name), name
    # LBM.h:32-56,97-118
    assert (p.nx, p.ny, p.nz, p.n_lattices, p.pb_iterations) == (50, 8, 51, 4, 501)
    assert p.dx == 1.0e-6 / 100.0 and p.CFL == 0.01 and p.rho0 == 1000.0
    assert p.cs_square == 1.0 / 3.0 / (0.01 * 0.01)
    assert p.voltage == -5.2574e-3 and p.K == 4.245e-7 and p.Kn == -4.245e-7
    assert abs(p.Lz - 0.5e-6) < 1e-20 and abs(p.Lx - 0.5e-6) < 1e-20


def test_params_struct_size_matches_c(pkg):
    # 6 int32 + 34 doubles, no padding surprises
    assert C.sizeof(pkg.Params) == 6 * 4 + 34 * 8


def test_invalid_arguments_return_errors_not_exit(pkg):
    lib = pkg.load_library()
    p = pkg.default_params(8, 8, 8)
    h = C.c_void_p()
    p.n_lattices = 2
    assert lib.ekpnp_create(C.byref(p), C.byref(h)) == 1
    assert b"n_lattices" in lib.ekpnp_last_error(None)
    p.n_lattices = 3  # Ra != 0 with 3 lattices is not parity-safe
    assert lib.ekpnp_create(C.byref(p), C.byref(h)) == 1
    p = pkg.default_params(8, 8, 7)
    assert lib.ekpnp_create_slab(C.byref(p), 0, 2, C.byref(h)) == 1  # slabs of 3 and 4 planes: too thin
    assert b"4 planes" in lib.ekpnp_last_error(None)
    p = pkg.default_params(8, 70000, 8)
    assert lib.ekpnp_create(C.byref(p), C.byref(h)) == 1
    assert b"ny" in lib.ekpnp_last_error(None)
    assert lib.ekpnp_step(None, 1) == 1
    assert lib.ekpnp_destroy(None) == 1


@pytest.mark.skipif(_has_gpu(), reason="only meaningful on a box without a HIP device")
def test_create_fails_loudly_without_gpu(pkg):
    p = pkg.default_params(8, 8, 8)
    with pytest.raises(pkg.EkpnpError):
        pkg.Solver(p)


def test_missing_library_raises(pkg, monkeypatch, tmp_path):
    import ek_pnp_3d_amd.solver as S

    monkeypatch.setattr(S, "_lib", None)
    monkeypatch.setattr(S, "_HERE", str(tmp_path))
    with pytest.raises(S.EkpnpError):
        S.load_library()


def test_product_never_touches_oracle():
    """The product path must not import, link or call anything under oracle/."""
    pk = os.path.join(ROOT, "ek-pnp-3d_amd")
    for dirpath, _, files in os.walk(pk):
        if os.path.basename(dirpath) == "build":
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"oracle", txt, re.I), f"{f} mentions the oracle"


def test_compute_parameters_matches_the_reference_formulas(pkg):
    """compute_parameters, LBM.cu:2440-2444 (host arithmetic: runs without a GPU)."""
    import math

    p = pkg.default_params(50, 8, 51)
    d = pkg.compute_parameters(p)
    assert d["M"] == math.sqrt(p.eps / p.rho0) / p.K
    assert d["T"] == p.eps * p.voltage / p.K / p.nu / p.rho0
    assert d["C"] == p.chargeinf * p.Lz * p.Lz / (p.voltage * p.eps)
    assert d["Fe"] == p.K * p.voltage / p.diffu
    assert d["Pr"] == p.nu / p.D == 1.0


def test_missing_rccl_is_an_error_with_a_message_not_a_crash(tmp_path):
    """librccl is bound on first use (dlopen).  When it cannot be loaded - EKPNP_RCCL_LIBRARY points the loader at a file
    that does not exist - ekpnp_comm_unique_id must return EKPNP_ERR_HIP and leave the loader's message in
    ekpnp_last_error(NULL); the round-2 code read dlerror() twice and built a std::string from NULL.  A child process:
    the binding is attempted once per process.  No GPU needed."""
    import subprocess
    import sys

    code = r'''
import ctypes as C, sys
sys.path.insert(0, %r)
import __graft_entry__ as G
pkg = G.load_package()
L = pkg.load_library()
buf = C.create_string_buffer(128)
rc = L.ekpnp_comm_unique_id(buf)
print("RC", rc, "MSG", L.ekpnp_last_error(None).decode())
rc2 = L.ekpnp_comm_unique_id(buf)   # asked again: same answer, still no crash
print("RC2", rc2)
''' % ROOT
    env = dict(os.environ, EKPNP_RCCL_LIBRARY=str(tmp_path / "no_such_librccl.so"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr[-2000:])
    assert "RC 2 MSG ekpnp_comm_unique_id:" in r.stdout and "no_such_librccl.so cannot be loaded" in r.stdout, r.stdout
    assert "RC2 2" in r.stdout
