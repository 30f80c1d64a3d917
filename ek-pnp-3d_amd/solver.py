"""ctypes binding of libekpnp.so (include/ekpnp.h).

`Solver` mirrors the reference's host step API one to one:

    reference (LBM.h)                      here
    -------------------------------------  ------------------------------
    initialization(r,c,cn,fi,u,v,w,...)    Solver.initialization()
    init_equilibrium(f0,f1,...,temp)       Solver.init_equilibrium()
    stream_collide_save(f0,...,t,f0bc)     Solver.stream_collide_save(t)
    fast_Poisson(charge,chargen,kx,ky,kz)  Solver.fast_Poisson()
    main.cu:189-200 loop body x n          Solver.step(n)

Errors: the reference prints and exit()s (LBM.cu:35-53); here every non-zero status of the
C ABI raises EkpnpError carrying ekpnp_last_error().
"""
from __future__ import annotations

import ctypes as C
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBNAME = "libekpnp.so"

FIELDS = ["rho", "c", "cn", "phi", "ux", "uy", "uz", "Ex", "Ey", "Ez", "T"]
FIELD_ID = {n: i for i, n in enumerate(FIELDS)}


class EkpnpError(RuntimeError):
    pass


class Params(C.Structure):
    """Mirror of `ekpnp_params` (include/ekpnp.h)."""

    _fields_ = [(n, C.c_int32) for n in ("nx", "ny", "nz", "n_lattices", "pb_iterations", "in_place")] + [
        (n, C.c_double)
        for n in (
            "Lx Ly Lz dx dy dz CFL dt cs_square rho0 chargeinf voltage voltage2 Ext eps "
            "diffu diffun nu K Kn D Ra TH uw exf kB electron roomT convertCtoCharge "
            "PB_omega V VC VCn VT"
        ).split()
    ]

    def copy(self) -> "Params":
        q = Params()
        C.memmove(C.byref(q), C.byref(self), C.sizeof(Params))
        return q


def library_path() -> str:
    # EKPNP_LIBRARY: A/B a differently built libekpnp.so (tools/sweep.sh); never a fallback
    return os.environ.get("EKPNP_LIBRARY") or os.path.join(_HERE, _LIBNAME)


def exported_symbols() -> list:
    """Names declared `int|size_t|const char* ekpnp_*(` in include/ekpnp.h."""
    hdr = os.path.join(_HERE, "..", "include", "ekpnp.h")
    txt = open(hdr).read()
    return sorted(set(re.findall(r"\b(ekpnp_[a-z0-9_]+)\s*\(", txt)))


_lib = None


def load_library():
    """dlopen libekpnp.so; raises EkpnpError if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise EkpnpError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback."
        )
    try:
        # PyTorch-ROCm is the plumbing for device memory / streams / torch.distributed.  It ships
        # its own HIP runtime + rocFFT; load it first so that one runtime serves the process.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    ctx = C.c_void_p
    i32, dbl, sz = C.c_int, C.c_double, C.c_size_t
    pd = C.POINTER(C.c_double)
    sig = {
        "ekpnp_default_params": (i32, [C.POINTER(Params), i32, i32, i32]),
        "ekpnp_create": (i32, [C.POINTER(Params), C.POINTER(ctx)]),
        "ekpnp_create_slab": (i32, [C.POINTER(Params), i32, i32, C.POINTER(ctx)]),
        "ekpnp_destroy": (i32, [ctx]),
        "ekpnp_last_error": (C.c_char_p, [ctx]),
        "ekpnp_set_stream": (i32, [ctx, C.c_void_p]),
        "ekpnp_synchronize": (i32, [ctx]),
        "ekpnp_bind_field": (i32, [ctx, i32, C.c_void_p]),
        "ekpnp_field_device_ptr": (i32, [ctx, i32, C.POINTER(C.c_void_p)]),
        "ekpnp_set_field": (i32, [ctx, i32, C.c_void_p]),
        "ekpnp_get_field": (i32, [ctx, i32, C.c_void_p]),
        "ekpnp_initialization": (i32, [ctx]),
        "ekpnp_initialization_converged": (i32, [ctx, dbl, i32, C.POINTER(i32), pd]),
        "ekpnp_init_equilibrium": (i32, [ctx]),
        "ekpnp_stream_collide_save": (i32, [ctx, dbl]),
        "ekpnp_fast_poisson": (i32, [ctx]),
        "ekpnp_invalidate_rhs": (i32, [ctx]),
        "ekpnp_step": (i32, [ctx, i32]),
        "ekpnp_get_time": (i32, [ctx, pd]),
        "ekpnp_set_time": (i32, [ctx, dbl]),
        "ekpnp_local_extent": (i32, [ctx, C.POINTER(i32), C.POINTER(i32)]),
        "ekpnp_kernel_timing_enable": (i32, [ctx, i32]),
        "ekpnp_kernel_timing_get": (i32, [ctx, C.POINTER(i32), pd, C.POINTER(C.c_int64)]),
        "ekpnp_phase_timing_get": (i32, [ctx, C.POINTER(i32), pd]),
        "ekpnp_poisson_stage_timing_get": (i32, [ctx, C.POINTER(i32), pd]),
        "ekpnp_plane_transforms": (i32, [ctx, C.POINTER(i32), C.POINTER(i32)]),
        "ekpnp_pass_order": (i32, [ctx, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
        "ekpnp_device_bytes": (sz, [ctx]),
        "ekpnp_placement_report": (i32, [ctx, C.POINTER(i32), C.POINTER(i32), pd, i32]),
        "ekpnp_graph_state": (i32, [ctx]),
        "ekpnp_debug_sync_enabled": (i32, []),
        "ekpnp_tune": (i32, [ctx, C.c_char_p, i32]),
        "ekpnp_copy_bandwidth": (i32, [ctx, sz, pd]),
        "ekpnp_halo_buffer": (i32, [ctx, i32, C.POINTER(C.c_void_p), C.POINTER(sz)]),
        "ekpnp_halo_pack": (i32, [ctx]),
        "ekpnp_halo_unpack": (i32, [ctx]),
        "ekpnp_phi_halo_buffer": (i32, [ctx, i32, C.POINTER(C.c_void_p), C.POINTER(sz)]),
        "ekpnp_poisson_stage1": (i32, [ctx]),
        "ekpnp_poisson_edge_buffer": (i32, [ctx, i32, C.POINTER(C.c_void_p), C.POINTER(sz)]),
        "ekpnp_poisson_stage2": (i32, [ctx]),
        "ekpnp_phi_halo_pack": (i32, [ctx]),
        "ekpnp_poisson_stage3": (i32, [ctx]),
        "ekpnp_collide_boundary_planes": (i32, [ctx]),
        "ekpnp_collide_interior_planes": (i32, [ctx]),
        "ekpnp_init_fields": (i32, [ctx]),
        "ekpnp_pbe_begin": (i32, [ctx]),
        "ekpnp_pbe_concentrations": (i32, [ctx]),
        "ekpnp_pbe_relax": (i32, [ctx]),
        "ekpnp_pbe_end": (i32, [ctx]),
        "ekpnp_advance_time": (i32, [ctx]),
        "ekpnp_current": (i32, [ctx, pd]),
        "ekpnp_umax": (i32, [ctx, pd]),
        "ekpnp_record_umax": (i32, [ctx, C.c_char_p, i32, dbl]),
        "ekpnp_save_data_tecplot": (i32, [ctx, C.c_char_p, i32, dbl, i32]),
        "ekpnp_save_data_end": (i32, [ctx, C.c_char_p, i32, dbl]),
        "ekpnp_read_data": (i32, [ctx, C.c_char_p, pd]),
        "ekpnp_save_state": (i32, [ctx, C.c_char_p, dbl]),
        "ekpnp_read_state": (i32, [ctx, C.c_char_p, pd]),
        "ekpnp_save_checkpoint": (i32, [ctx, C.c_char_p]),
        "ekpnp_load_checkpoint": (i32, [ctx, C.c_char_p, pd]),
        "ekpnp_group_save_checkpoint": (i32, [ctx, C.c_char_p]),
        "ekpnp_group_load_checkpoint": (i32, [ctx, C.c_char_p, pd]),
        "ekpnp_compute_parameters": (i32, [C.POINTER(Params), pd, pd, pd, pd, pd]),
        "ekpnp_save_scalar": (i32, [ctx, C.c_char_p, i32, C.c_uint, C.c_uint]),
        # the library's own halo transport (RCCL / peer copies)
        "ekpnp_comm_unique_id": (i32, [C.c_void_p]),
        "ekpnp_rccl_available": (i32, []),
        "ekpnp_slab_attach_comm": (i32, [ctx, C.c_void_p]),
        "ekpnp_comm_timing_get": (i32, [ctx, i32, C.POINTER(i32), pd, pd, C.POINTER(sz)]),
        "ekpnp_group_create": (i32, [C.POINTER(Params), i32, C.POINTER(i32), i32, C.POINTER(ctx)]),
        "ekpnp_group_destroy": (i32, [ctx]),
        "ekpnp_group_last_error": (C.c_char_p, [ctx]),
        "ekpnp_group_size": (i32, [ctx]),
        "ekpnp_group_transport": (i32, [ctx]),
        "ekpnp_group_context": (i32, [ctx, i32, C.POINTER(ctx)]),
        "ekpnp_group_device_bytes": (sz, [ctx]),
        "ekpnp_group_synchronize": (i32, [ctx]),
        "ekpnp_group_set_field": (i32, [ctx, i32, C.c_void_p]),
        "ekpnp_group_get_field": (i32, [ctx, i32, C.c_void_p]),
        "ekpnp_group_initialization": (i32, [ctx]),
        "ekpnp_group_initialization_converged": (i32, [ctx, dbl, i32, C.POINTER(i32), pd]),
        "ekpnp_group_init_equilibrium": (i32, [ctx]),
        "ekpnp_group_stream_collide_save": (i32, [ctx, dbl]),
        "ekpnp_group_fast_poisson": (i32, [ctx]),
        "ekpnp_group_step": (i32, [ctx, i32]),
        "ekpnp_group_tune": (i32, [ctx, C.c_char_p, i32]),
        "ekpnp_group_get_time": (i32, [ctx, pd]),
        "ekpnp_group_set_time": (i32, [ctx, dbl]),
        "ekpnp_group_current": (i32, [ctx, pd]),
        "ekpnp_group_umax": (i32, [ctx, pd]),
        "ekpnp_group_record_umax": (i32, [ctx, C.c_char_p, i32, dbl]),
        "ekpnp_group_save_data_tecplot": (i32, [ctx, C.c_char_p, i32, dbl, i32]),
        "ekpnp_group_save_data_end": (i32, [ctx, C.c_char_p, i32, dbl]),
        "ekpnp_group_read_data": (i32, [ctx, C.c_char_p, pd]),
        "ekpnp_group_save_state": (i32, [ctx, C.c_char_p, dbl]),
        "ekpnp_group_read_state": (i32, [ctx, C.c_char_p, pd]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def default_params(nx: int, ny: int, nz: int) -> Params:
    p = Params()
    rc = load_library().ekpnp_default_params(C.byref(p), nx, ny, nz)
    if rc:
        raise EkpnpError(f"ekpnp_default_params({nx},{ny},{nz}) -> {rc}")
    return p


def slab_extent(nz: int, rank: int, nranks: int):
    """(first plane, number of planes) owned by `rank`: planes [rank*nz//nranks, (rank+1)*nz//nranks),
    the same rule as ekpnp_create_slab (slabs differ by at most one plane)."""
    if nranks > 1 and nz // nranks < 4:
        raise ValueError("each z slab needs at least 4 planes")
    z0 = rank * nz // nranks
    return z0, (rank + 1) * nz // nranks - z0


def compute_parameters(p: Params) -> dict:
    """compute_parameters (LBM.cu:2419-2446): the dimensionless groups T, M, C, Fe, Pr."""
    v = [C.c_double() for _ in range(5)]
    rc = load_library().ekpnp_compute_parameters(C.byref(p), *[C.byref(x) for x in v])
    if rc:
        raise EkpnpError(f"ekpnp_compute_parameters -> {rc}")
    return dict(zip(("T", "M", "C", "Fe", "Pr"), (x.value for x in v)))


TRANSPORT_AUTO, TRANSPORT_RCCL, TRANSPORT_COPY = 0, 1, 2
UNIQUE_ID_BYTES = 128


def comm_unique_id() -> bytes:
    """ncclGetUniqueId through the library: made by ONE rank, handed to all (ekpnp_comm_unique_id)."""
    buf = C.create_string_buffer(UNIQUE_ID_BYTES)
    rc = load_library().ekpnp_comm_unique_id(buf)
    if rc:
        raise EkpnpError(f"ekpnp_comm_unique_id -> {rc} (librccl.so.1 not loadable?)")
    return buf.raw


def rccl_available() -> str:
    """'' when this process can bind the RCCL library the transport uses (ekpnp_rccl_available: no device, no
    communicator), else the loader's message.  Hosts call it on every rank and agree BEFORE attaching."""
    L = load_library()
    if L.ekpnp_rccl_available() == 0:
        return ""
    return (L.ekpnp_last_error(None) or b"RCCL cannot be bound").decode()


COMM_KINDS = ("halo", "phi", "edge")  # ekpnp_comm_timing_get kinds 0, 1, 2
STAGE_NAMES = ("stage1", "edge_exchange", "stage2", "phi_exchange", "stage3")  # ekpnp_poisson_stage_timing_get


def comm_timing(L, handle, check) -> dict:
    out = {}
    for kind, name in enumerate(COMM_KINDS):
        n, w, t, b = C.c_int(), C.c_double(), C.c_double(), C.c_size_t()
        check(L.ekpnp_comm_timing_get(handle, kind, C.byref(n), C.byref(w), C.byref(t), C.byref(b)))
        out[name] = {"n": n.value, "wait_ms": w.value, "transfer_ms": t.value, "bytes_sent": int(b.value)}
    return out


class Solver:
    """One EK-PNP simulation on the current HIP device (or one z slab of it)."""

    def __init__(self, params: Params, rank: int = 0, nranks: int = 1, slab: bool = False):
        self._L = load_library()
        self.p = params.copy()
        self._h = C.c_void_p()
        if nranks == 1 and not slab:
            rc = self._L.ekpnp_create(C.byref(self.p), C.byref(self._h))
        else:
            rc = self._L.ekpnp_create_slab(C.byref(self.p), rank, nranks, C.byref(self._h))
        if rc:
            msg = self._L.ekpnp_last_error(None).decode()
            self._h = C.c_void_p()
            raise EkpnpError(f"ekpnp_create failed ({rc}): {msg}")
        z0, nzl = C.c_int(), C.c_int()
        self._ck(self._L.ekpnp_local_extent(self._h, C.byref(z0), C.byref(nzl)))
        self.z0, self.nz_local = z0.value, nzl.value
        self.shape = (self.nz_local, self.p.ny, self.p.nx)
        self.rank, self.nranks = rank, nranks

    # -- plumbing ---------------------------------------------------------------------------
    def _ck(self, rc: int):
        if rc:
            raise EkpnpError(f"status {rc}: {self._L.ekpnp_last_error(self._h).decode()}")

    def close(self):
        if getattr(self, "_h", None):
            self._L.ekpnp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def handle(self):
        return self._h

    @property
    def lib(self):
        return self._L

    def synchronize(self):
        self._ck(self._L.ekpnp_synchronize(self._h))

    def set_stream(self, hip_stream: int):
        self._ck(self._L.ekpnp_set_stream(self._h, C.c_void_p(hip_stream)))

    def graph_state(self) -> int:
        return int(self._L.ekpnp_graph_state(self._h))

    def device_bytes(self) -> int:
        return int(self._L.ekpnp_device_bytes(self._h))

    def placement_report(self) -> dict:
        """{"tried": n, "chosen": i, "sweep_ms": [...]}: the arenas ekpnp_create timed and the one it kept (tried == 0: no search)"""
        n, ch, ms = C.c_int(), C.c_int(), (C.c_double * 8)()
        self._ck(self._L.ekpnp_placement_report(self._h, C.byref(n), C.byref(ch), ms, 8))
        return {"tried": n.value, "chosen": ch.value, "sweep_ms": [round(ms[k], 4) for k in range(n.value)]}

    def copy_bandwidth(self, nbytes: int = 1 << 32) -> float:
        """GB/s (read + write) of a plain contiguous device copy: the measured streaming ceiling."""
        v = C.c_double()
        self._ck(self._L.ekpnp_copy_bandwidth(self._h, int(nbytes), C.byref(v)))
        return v.value

    # -- fields -----------------------------------------------------------------------------
    def get_field(self, name: str) -> np.ndarray:
        out = np.empty(self.shape, dtype=np.float64)
        self._ck(self._L.ekpnp_get_field(self._h, FIELD_ID[name], out.ctypes.data_as(C.c_void_p)))
        return out

    def set_field(self, name: str, value):
        a = np.ascontiguousarray(value, dtype=np.float64).reshape(self.shape)
        self._ck(self._L.ekpnp_set_field(self._h, FIELD_ID[name], a.ctypes.data_as(C.c_void_p)))

    def fields(self) -> dict:
        return {n: self.get_field(n) for n in FIELDS}

    def set_fields(self, d: dict):
        for n, v in d.items():
            self.set_field(n, v)

    def field_device_ptr(self, name: str) -> int:
        p = C.c_void_p()
        self._ck(self._L.ekpnp_field_device_ptr(self._h, FIELD_ID[name], C.byref(p)))
        return int(p.value)

    def bind_field(self, name: str, device_ptr: int):
        self._ck(self._L.ekpnp_bind_field(self._h, FIELD_ID[name], C.c_void_p(device_ptr)))

    # -- the reference's host API (LBM.h:159-180) ---------------------------------------------
    def initialization(self):
        self._ck(self._L.ekpnp_initialization(self._h))

    def initialization_converged(self, rel_tol: float = 1e-10, max_sweeps: int = 100000):
        """initialization() with a convergence test; returns (sweeps done, relative residual)."""
        n, r = C.c_int(), C.c_double()
        self._ck(self._L.ekpnp_initialization_converged(self._h, float(rel_tol), int(max_sweeps), C.byref(n), C.byref(r)))
        return n.value, r.value

    def init_equilibrium(self):
        self._ck(self._L.ekpnp_init_equilibrium(self._h))

    def stream_collide_save(self, t: float = 0.0):
        self._ck(self._L.ekpnp_stream_collide_save(self._h, float(t)))

    def fast_Poisson(self):
        self._ck(self._L.ekpnp_fast_poisson(self._h))

    def step(self, n: int = 1):
        self._ck(self._L.ekpnp_step(self._h, int(n)))

    @property
    def t(self) -> float:
        v = C.c_double()
        self._ck(self._L.ekpnp_get_time(self._h, C.byref(v)))
        return v.value

    # -- diagnostics and IO (LBM.cu:2492-2753) -----------------------------------------------
    def current(self) -> float:
        v = C.c_double()
        self._ck(self._L.ekpnp_current(self._h, C.byref(v)))
        return v.value

    def umax(self) -> float:
        v = C.c_double()
        self._ck(self._L.ekpnp_umax(self._h, C.byref(v)))
        return v.value

    def record_umax(self, path: str, time: float, append: bool = True):
        self._ck(self._L.ekpnp_record_umax(self._h, os.fsencode(path), int(append), float(time)))

    def save_data_tecplot(self, path: str, time: float, first: bool = True, append: bool = False):
        self._ck(self._L.ekpnp_save_data_tecplot(self._h, os.fsencode(path), int(append), float(time), int(first)))

    def save_data_end(self, path: str, time: float, append: bool = False):
        self._ck(self._L.ekpnp_save_data_end(self._h, os.fsencode(path), int(append), float(time)))

    def save_scalar(self, name: str, field: str, n: int, nsteps: int = 1000):
        self._ck(self._L.ekpnp_save_scalar(self._h, os.fsencode(name), FIELD_ID[field], int(n), int(nsteps)))

    def read_data(self, path: str) -> float:
        t = C.c_double()
        self._ck(self._L.ekpnp_read_data(self._h, os.fsencode(path), C.byref(t)))
        return t.value

    def save_state(self, path: str, time: float = 0.0):
        """Lossless binary variant of save_data_end (raw FP64 fields of the owned planes)."""
        self._ck(self._L.ekpnp_save_state(self._h, os.fsencode(path), float(time)))

    def read_state(self, path: str) -> float:
        t = C.c_double()
        self._ck(self._L.ekpnp_read_state(self._h, os.fsencode(path), C.byref(t)))
        return t.value

    def tune(self, knob: str, value: int):
        self._ck(self._L.ekpnp_tune(self._h, knob.encode(), int(value)))

    def invalidate_rhs(self):
        self._ck(self._L.ekpnp_invalidate_rhs(self._h))

    def attach_comm(self, unique_id: bytes):
        """Collective over the ranks (ncclCommInitRank): from here on the reference's verbs
        (initialization, stream_collide_save, fast_Poisson, step, current, umax, the writers)
        work on this slab context, the library moving the halos over RCCL itself."""
        if len(unique_id) != UNIQUE_ID_BYTES:
            raise ValueError("unique id must be 128 bytes")
        self._ck(self._L.ekpnp_slab_attach_comm(self._h, C.c_char_p(unique_id)))

    def save_checkpoint(self, path: str):
        """fields + post-collision populations: load_checkpoint continues the run bit for bit"""
        self._ck(self._L.ekpnp_save_checkpoint(self._h, os.fsencode(path)))

    def load_checkpoint(self, path: str) -> float:
        t = C.c_double()
        self._ck(self._L.ekpnp_load_checkpoint(self._h, os.fsencode(path), C.byref(t)))
        return t.value

    # -- z-slab pieces: the split entry points (a host's own transport, examples/host_transport.py; tests) ------
    def call(self, name: str):
        """Invoke a parameterless `int ekpnp_<name>(ctx)` entry point."""
        self._ck(getattr(self._L, "ekpnp_" + name)(self._h))

    def buffer(self, kind: str, which: int):
        """(device pointer, n_doubles) of a halo / phi-halo / edge buffer."""
        p, n = C.c_void_p(), C.c_size_t()
        fn = {"halo": self._L.ekpnp_halo_buffer, "phi": self._L.ekpnp_phi_halo_buffer, "edge": self._L.ekpnp_poisson_edge_buffer}[kind]
        self._ck(fn(self._h, which, C.byref(p), C.byref(n)))
        return int(p.value), int(n.value)

    # -- measurement ------------------------------------------------------------------------
    def kernel_timing(self, enable: bool):
        self._ck(self._L.ekpnp_kernel_timing_enable(self._h, int(enable)))

    def phase_timing_get(self):
        """(number of Poisson solves bracketed since the last call, their summed duration in ms)"""
        n, ms = C.c_int(), C.c_double()
        self._ck(self._L.ekpnp_phase_timing_get(self._h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def plane_transforms(self) -> dict:
        """{"own_passes": the library's own row / column kernels (else rocFFT plans), "ranks_on_device": ranks of the lattice
        sharing this context's device (known once a communicator is attached)}"""
        own, n = C.c_int(), C.c_int()
        self._ck(self._L.ekpnp_plane_transforms(self._h, C.byref(own), C.byref(n)))
        return {"own_passes": bool(own.value), "ranks_on_device": n.value}

    def pass_order(self) -> dict:
        """the cache-aware orders in effect: {"band_rows": rows per band of the interior sweep (0: plane after plane),
        "poisson_blocks": kx column blocks of the solve's middle passes, "poisson_zchunk": planes per chunk of its row + column passes}"""
        b, nb, zc = C.c_int(), C.c_int(), C.c_int()
        self._ck(self._L.ekpnp_pass_order(self._h, C.byref(b), C.byref(nb), C.byref(zc)))
        return {"band_rows": b.value, "poisson_blocks": nb.value, "poisson_zchunk": zc.value}

    def poisson_stage_timing_get(self):
        """Slab contexts: (n solves, {"stage1", "edge_exchange", "stage2", "phi_exchange", "stage3"}: summed ms) of the
        solves bracketed so far - call BEFORE phase_timing_get, which resets them."""
        n, ms = C.c_int(), (C.c_double * 5)()
        self._ck(self._L.ekpnp_poisson_stage_timing_get(self._h, C.byref(n), ms))
        return n.value, dict(zip(STAGE_NAMES, (float(v) for v in ms)))

    def kernel_timing_get(self):
        n, ms, nodes = C.c_int(), C.c_double(), C.c_int64()
        self._ck(self._L.ekpnp_kernel_timing_get(self._h, C.byref(n), C.byref(ms), C.byref(nodes)))
        return n.value, ms.value, nodes.value

    def comm_timing_get(self) -> dict:
        """Per exchange kind of a slab with a transport, since kernel_timing(True): {"halo"|"phi"|"edge":
        {"n", "wait_ms", "transfer_ms", "bytes_sent"}} - sums over the n exchanges, bytes per exchange."""
        return comm_timing(self._L, self._h, self._ck)


class Group:
    """nslabs z slabs driven by ONE process (ekpnp_group_*): slab i on HIP device devices[i]
    (default: i modulo the device count; devices may repeat, then the halos move by device copies).
    Same method names as Solver; fields are whole-lattice arrays [NZ][NY][NX]."""

    def __init__(self, params: Params, nslabs: int, devices=None, transport: int = TRANSPORT_AUTO):
        self._L = load_library()
        self.p = params.copy()
        self._g = C.c_void_p()
        dev = None
        if devices is not None:
            if len(devices) != nslabs:
                raise ValueError("one device per slab")
            dev = (C.c_int * nslabs)(*devices)
        rc = self._L.ekpnp_group_create(C.byref(self.p), int(nslabs), dev, int(transport), C.byref(self._g))
        if rc:
            msg = self._L.ekpnp_group_last_error(None).decode()
            self._g = C.c_void_p()
            raise EkpnpError(f"ekpnp_group_create failed ({rc}): {msg}")
        self.n = nslabs
        self.shape = (self.p.nz, self.p.ny, self.p.nx)
        self.transport = int(self._L.ekpnp_group_transport(self._g))

    def _ck(self, rc: int):
        if rc:
            raise EkpnpError(f"status {rc}: {self._L.ekpnp_group_last_error(self._g).decode()}")

    def close(self):
        if getattr(self, "_g", None):
            self._L.ekpnp_group_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def slab_handle(self, i: int):
        h = C.c_void_p()
        self._ck(self._L.ekpnp_group_context(self._g, int(i), C.byref(h)))
        return h

    def slab_extent(self, i: int):
        z0, nzl = C.c_int(), C.c_int()
        rc = self._L.ekpnp_local_extent(self.slab_handle(i), C.byref(z0), C.byref(nzl))
        if rc:
            raise EkpnpError(f"ekpnp_local_extent -> {rc}")
        return z0.value, nzl.value

    def device_bytes(self) -> int:
        return int(self._L.ekpnp_group_device_bytes(self._g))

    # measurement: the hooks are per slab context
    def kernel_timing(self, enable: bool):
        for i in range(self.n):
            self._slab_ck(i, self._L.ekpnp_kernel_timing_enable(self.slab_handle(i), int(enable)))

    def slab_kernel_timing_get(self, i: int):
        n, ms, nodes = C.c_int(), C.c_double(), C.c_int64()
        self._slab_ck(i, self._L.ekpnp_kernel_timing_get(self.slab_handle(i), C.byref(n), C.byref(ms), C.byref(nodes)))
        return n.value, ms.value, nodes.value

    def slab_phase_timing_get(self, i: int):
        n, ms = C.c_int(), C.c_double()
        self._slab_ck(i, self._L.ekpnp_phase_timing_get(self.slab_handle(i), C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def slab_comm_timing_get(self, i: int) -> dict:
        h = self.slab_handle(i)
        return comm_timing(self._L, h, lambda rc: self._slab_ck(i, rc))

    def _slab_ck(self, i: int, rc: int):
        if rc:
            raise EkpnpError(f"slab {i}: status {rc}: {self._L.ekpnp_last_error(self.slab_handle(i)).decode()}")

    def synchronize(self):
        self._ck(self._L.ekpnp_group_synchronize(self._g))

    def get_field(self, name: str) -> np.ndarray:
        out = np.empty(self.shape, dtype=np.float64)
        self._ck(self._L.ekpnp_group_get_field(self._g, FIELD_ID[name], out.ctypes.data_as(C.c_void_p)))
        return out

    def set_field(self, name: str, value):
        a = np.ascontiguousarray(value, dtype=np.float64).reshape(self.shape)
        self._ck(self._L.ekpnp_group_set_field(self._g, FIELD_ID[name], a.ctypes.data_as(C.c_void_p)))

    def fields(self) -> dict:
        return {n: self.get_field(n) for n in FIELDS}

    def set_fields(self, d: dict):
        for n, v in d.items():
            self.set_field(n, v)

    def initialization(self):
        self._ck(self._L.ekpnp_group_initialization(self._g))

    def initialization_converged(self, rel_tol: float = 1e-10, max_sweeps: int = 100000):
        n, r = C.c_int(), C.c_double()
        self._ck(self._L.ekpnp_group_initialization_converged(self._g, float(rel_tol), int(max_sweeps), C.byref(n), C.byref(r)))
        return n.value, r.value

    def init_equilibrium(self):
        self._ck(self._L.ekpnp_group_init_equilibrium(self._g))

    def stream_collide_save(self, t: float = 0.0):
        self._ck(self._L.ekpnp_group_stream_collide_save(self._g, float(t)))

    def fast_Poisson(self):
        self._ck(self._L.ekpnp_group_fast_poisson(self._g))

    def tune(self, knob: str, value: int):
        """ekpnp_tune's slab and transport knobs on every slab of the group"""
        self._ck(self._L.ekpnp_group_tune(self._g, knob.encode(), int(value)))

    def step(self, n: int = 1):
        self._ck(self._L.ekpnp_group_step(self._g, int(n)))

    @property
    def t(self) -> float:
        v = C.c_double()
        self._ck(self._L.ekpnp_group_get_time(self._g, C.byref(v)))
        return v.value

    def current(self) -> float:
        v = C.c_double()
        self._ck(self._L.ekpnp_group_current(self._g, C.byref(v)))
        return v.value

    def umax(self) -> float:
        v = C.c_double()
        self._ck(self._L.ekpnp_group_umax(self._g, C.byref(v)))
        return v.value

    def record_umax(self, path: str, time: float, append: bool = True):
        self._ck(self._L.ekpnp_group_record_umax(self._g, os.fsencode(path), int(append), float(time)))

    def save_data_tecplot(self, path: str, time: float, first: bool = True, append: bool = False):
        self._ck(self._L.ekpnp_group_save_data_tecplot(self._g, os.fsencode(path), int(append), float(time), int(first)))

    def save_data_end(self, path: str, time: float, append: bool = False):
        self._ck(self._L.ekpnp_group_save_data_end(self._g, os.fsencode(path), int(append), float(time)))

    def read_data(self, path: str) -> float:
        t = C.c_double()
        self._ck(self._L.ekpnp_group_read_data(self._g, os.fsencode(path), C.byref(t)))
        return t.value

    def save_state(self, path: str, time: float = 0.0):
        self._ck(self._L.ekpnp_group_save_state(self._g, os.fsencode(path), float(time)))

    def read_state(self, path: str) -> float:
        t = C.c_double()
        self._ck(self._L.ekpnp_group_read_state(self._g, os.fsencode(path), C.byref(t)))
        return t.value

    def save_checkpoint(self, path: str):
        self._ck(self._L.ekpnp_group_save_checkpoint(self._g, os.fsencode(path)))

    def load_checkpoint(self, path: str) -> float:
        t = C.c_double()
        self._ck(self._L.ekpnp_group_load_checkpoint(self._g, os.fsencode(path), C.byref(t)))
        return t.value
