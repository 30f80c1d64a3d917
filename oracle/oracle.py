"""ctypes front-end of the CPU oracle (oracle/ekpnp_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package never does.  The class mirrors the reference's host
functions one to one (LBM.h:159-180): initialization, init_equilibrium,
stream_collide_save, fast_Poisson, plus the four LBM sub-kernels for bisecting.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libekpnp_oracle.so")

FIELDS = ["rho", "c", "cn", "phi", "ux", "uy", "uz", "Ex", "Ey", "Ez", "T"]
FIELD_ID = {n: i for i, n in enumerate(FIELDS)}
LATTICES = ["f", "h", "hn", "temp"]


class Params(C.Structure):
    """Mirror of `ekpnp_params` (include/ekpnp.h)."""

    _fields_ = (
        [(n, C.c_int32) for n in ("nx", "ny", "nz", "n_lattices", "pb_iterations", "in_place")]
        + [
            (n, C.c_double)
            for n in (
                "Lx Ly Lz dx dy dz CFL dt cs_square rho0 chargeinf voltage voltage2 Ext eps "
                "diffu diffun nu K Kn D Ra TH uw exf kB electron roomT convertCtoCharge "
                "PB_omega V VC VCn VT"
            ).split()
        ]
    )

    def copy(self) -> "Params":
        q = Params()
        C.memmove(C.byref(q), C.byref(self), C.sizeof(Params))
        return q

    def as_dict(self) -> dict:
        return {n: getattr(self, n) for n, _ in self._fields_}


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "ekpnp_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        # EKPNP_ORACLE_LIBRARY: another build of the same source - the AddressSanitizer / UBSan build of `make asan`
        # (tests/test_oracle_cpu.py::test_oracle_under_sanitizers runs a child process with it)
        named = os.environ.get("EKPNP_ORACLE_LIBRARY")
        L = C.CDLL(named if named else build())
        L.oracle_default_params.argtypes = [C.POINTER(Params), C.c_int, C.c_int, C.c_int]
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.POINTER(Params), C.c_int]
        L.oracle_destroy.argtypes = [C.c_void_p]
        L.oracle_field.restype = C.POINTER(C.c_double)
        L.oracle_field.argtypes = [C.c_void_p, C.c_int]
        L.oracle_population.restype = C.POINTER(C.c_double)
        L.oracle_population.argtypes = [C.c_void_p, C.c_int, C.c_int]
        for fn in (
            "oracle_initialization oracle_gpu_initialization oracle_gpu_PBE oracle_gpu_PBE_phi "
            "oracle_init_equilibrium oracle_collide_save oracle_boundary oracle_stream "
            "oracle_bc_charge oracle_stream_collide_save oracle_fast_poisson oracle_efield"
        ).split():
            getattr(L, fn).argtypes = [C.c_void_p]
            getattr(L, fn).restype = None
        L.oracle_step.argtypes = [C.c_void_p, C.c_int]
        L.oracle_step.restype = None
        L.oracle_set_dc_shift.argtypes = [C.c_void_p, C.c_double]
        L.oracle_set_dc_shift.restype = None
        for fn in ("oracle_initialization_shifts", "oracle_step_shifts"):
            getattr(L, fn).argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
            getattr(L, fn).restype = None
        for fn in ("oracle_current", "oracle_umax"):
            getattr(L, fn).argtypes = [C.c_void_p]
            getattr(L, fn).restype = C.c_double
        L.oracle_wall_velocity_alt.restype = C.POINTER(C.c_double)
        L.oracle_wall_velocity_alt.argtypes = [C.c_void_p]
        L.oracle_set_num_threads.argtypes = [C.c_int]
        L.oracle_set_num_threads.restype = None
        L.oracle_get_max_threads.restype = C.c_int
        _lib = L
    return _lib


def host_cores() -> int:
    """Cores the oracle may use: OMP_NUM_THREADS if set, else min(16, affinity) - the GPU box
    grants about 16 cores per GPU while exposing every hardware thread of the host."""
    if os.environ.get("OMP_NUM_THREADS"):
        return max(1, int(os.environ["OMP_NUM_THREADS"]))
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))


def set_threads(n: int) -> int:
    lib().oracle_set_num_threads(int(n))
    return lib().oracle_get_max_threads()


def default_params(nx: int, ny: int, nz: int) -> Params:
    p = Params()
    lib().oracle_default_params(C.byref(p), nx, ny, nz)
    return p


class Oracle:
    """One reference-layout simulation state on the CPU."""

    def __init__(self, params: Params, dc_mode: int = 0):
        self.p = params.copy()
        self._h = lib().oracle_create(C.byref(self.p), dc_mode)
        self.shape = (self.p.nz, self.p.ny, self.p.nx)
        self.n = self.p.nx * self.p.ny * self.p.nz
        # small lattices: a thread team costs more than it saves
        self.threads = set_threads(1 if self.n < 100_000 else host_cores())

    def close(self):
        if self._h:
            lib().oracle_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- views (no copies) -----------------------------------------------------------------
    def field(self, name: str) -> np.ndarray:
        ptr = lib().oracle_field(self._h, FIELD_ID[name])
        return np.ctypeslib.as_array(ptr, shape=self.shape)

    def fields(self) -> dict:
        return {n: self.field(n).copy() for n in FIELDS}

    def set_fields(self, d: dict):
        for n, v in d.items():
            self.field(n)[...] = np.asarray(v, dtype=np.float64).reshape(self.shape)

    def wall_velocity_alt(self) -> np.ndarray:
        """[8][3][NY][NX]: (ux, uy, uz) of plane z=0 under every resolution of the reference's
        read-after-write race (LBM.cu:664-667 vs 1711-1714): node z=1's rest population of h
        (mask bit 0), hn (bit 1), temp (bit 2) read AFTER its collision instead of before.  Mask 0
        is the canonical one the fields hold."""
        ptr = lib().oracle_wall_velocity_alt(self._h)
        return np.ctypeslib.as_array(ptr, shape=(8, 3, self.p.ny, self.p.nx))

    def population(self, lattice: str, which: int) -> np.ndarray:
        """which 0: rest X0[NZ][NY][NX]; 1: X1[26][NZ][NY][NX]; 2: X2 (LBM.cu:17-30)."""
        ptr = lib().oracle_population(self._h, LATTICES.index(lattice), which)
        shape = self.shape if which == 0 else (26,) + self.shape
        return np.ctypeslib.as_array(ptr, shape=shape)

    # -- the reference's host API ----------------------------------------------------------
    def initialization(self):
        lib().oracle_initialization(self._h)

    def init_equilibrium(self):
        lib().oracle_init_equilibrium(self._h)

    def stream_collide_save(self):
        lib().oracle_stream_collide_save(self._h)

    def fast_poisson(self, dc_shift: float = 0.0):
        """dc_shift: measured interior shift of one reference solve (see oracle_set_dc_shift)."""
        lib().oracle_set_dc_shift(self._h, float(dc_shift))
        lib().oracle_fast_poisson(self._h)
        lib().oracle_set_dc_shift(self._h, 0.0)

    def step(self, n: int = 1):
        lib().oracle_step(self._h, n)

    def step_shifts(self, shifts):
        a = np.ascontiguousarray(shifts, dtype=np.float64)
        lib().oracle_step_shifts(self._h, a.ctypes.data_as(C.POINTER(C.c_double)), a.size)

    def initialization_shifts(self, shifts):
        a = np.ascontiguousarray(shifts, dtype=np.float64)
        lib().oracle_initialization_shifts(self._h, a.ctypes.data_as(C.POINTER(C.c_double)), a.size)

    def current(self) -> float:
        return lib().oracle_current(self._h)

    def umax(self) -> float:
        return lib().oracle_umax(self._h)

    # -- sub-kernels -----------------------------------------------------------------------
    def collide_save(self):
        lib().oracle_collide_save(self._h)

    def boundary(self):
        lib().oracle_boundary(self._h)

    def stream(self):
        lib().oracle_stream(self._h)

    def bc_charge(self):
        lib().oracle_bc_charge(self._h)

    def gpu_initialization(self):
        lib().oracle_gpu_initialization(self._h)

    def efield(self):
        lib().oracle_efield(self._h)


def perturb_fields(p: Params, f: dict, rho_amp: float = 1e-6) -> dict:
    """Seed-free closed-form 3-D perturbation of SURVEY.md §8(c) applied on top of the
    fields left by `initialization` (c, cn, T, u, rho).  Returns new arrays."""
    nz, ny, nx = p.nz, p.ny, p.nx
    z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    X = 2 * np.pi * x / nx
    Y = 2 * np.pi * y / ny
    Z = np.pi * z / (nz - 1)
    g = {k: np.array(v, dtype=np.float64, copy=True) for k, v in f.items()}
    g["c"] = g["c"] * (1 + 0.02 * np.sin(X) * np.cos(2 * Y) * np.sin(Z))
    g["cn"] = g["cn"] * (1 + 0.02 * np.cos(2 * X) * np.sin(Y) * np.sin(Z))
    g["T"] = g["T"] + 0.05 * np.sin(X + Y) * np.sin(Z)
    g["ux"] = 1e-4 * np.sin(Z) * np.sin(X) * np.cos(Y)
    g["uy"] = -0.7e-4 * np.sin(Z) * np.cos(X) * np.sin(2 * Y)
    g["uz"] = 0.5e-4 * np.sin(Z) ** 2 * np.cos(X) * np.cos(Y)
    g["rho"] = g["rho"] * (1 + rho_amp * np.cos(X) * np.cos(Y) * np.sin(Z))
    return g


GROUPS = {
    "rho": ["rho"],
    "u": ["ux", "uy", "uz"],
    "c": ["c"],
    "cn": ["cn"],
    "phi": ["phi"],
    "T": ["T"],
    "E": ["Ex", "Ey", "Ez"],
}


def rel_l2(a: dict, b: dict, groups=GROUPS) -> dict:
    """rel-L2 per field group (vector fields jointly), SURVEY.md §8(c) comparison metric."""
    out = {}
    for g, names in groups.items():
        if not all(n in a and n in b for n in names):
            continue
        num = sum(float(np.sum((np.asarray(a[n], dtype=np.float64) - np.asarray(b[n], dtype=np.float64)) ** 2)) for n in names)
        den = sum(float(np.sum(np.asarray(b[n], dtype=np.float64) ** 2)) for n in names)
        out[g] = float(np.sqrt(num / den)) if den > 0 else float(np.sqrt(num))
    return out
