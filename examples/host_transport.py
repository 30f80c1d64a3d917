"""A HOST'S OWN TRANSPORT over the split entry points of libekpnp.so - the worked example of INTEGRATION.md section 5(c).

NOT part of the product package.  The library moves the halos of a z-slab run itself (csrc/slab_team.hip:
ekpnp_slab_attach_comm / ekpnp_group_*, RCCL or peer copies on a comm stream); that is the only data path the
package `ek-pnp-3d_amd/` has and the only one bench.py's headline lines run on.  This module shows how a host that
already owns a communication layer (here: torch.distributed) would drive the same slab contexts through
ekpnp_collide_boundary_planes / ekpnp_halo_pack / ... / ekpnp_poisson_stage1..3 and move the buffers itself.  It is
used by tests (gloo rehearsal of the multi-process flow on CPU / a one-GPU box: tests/_slab_worker.py,
tests/_ring_worker.py; `LocalSlabGroup`: every slab kernel in one process) and by bench.py only behind explicit flags
(`--backend gloo` rehearsals, `--allow-fallback-transport`).

No reference counterpart: the reference is single-GPU (`cudaSetDevice(0)`, main.cu:58).  One process per GPU; rank r
owns planes [r*NZ/P, (r+1)*NZ/P).  Per step a rank exchanges

  * LBM halo: the 9 c_z=+1 populations of its top plane go up, the 9 c_z=-1 of its bottom plane go down, per active
    lattice (9*L*8*NX*NY bytes per face).  `gpu_stream` wraps z (LBM.cu:1972,1975), so the neighbour graph is a RING.
  * Poisson: 4 interface coefficients per (kx,ky) mode, all-gathered, then every rank solves the same tiny interface
    system and corrects its own rows.
  * one phi plane each way for Ez (poisson.cu:50-55).

`RingTransport` is the only place that talks to `torch.distributed`; it works on any 1-D float64 tensors (CUDA tensors
over RCCL/"nccl", or CPU tensors over "gloo", staging through the host).  `LocalSlabGroup` runs P slab contexts inside
ONE process on ONE device with device-to-device copies as transport.

Import after the package has been loaded (`__graft_entry__.load_package()`), with the repository root on sys.path:
`from examples.host_transport import DistributedSlab`.
"""
from __future__ import annotations

import numpy as np

from ek_pnp_3d_amd import FIELDS, Params, Solver, slab_extent  # noqa: F401  (the package: __graft_entry__.load_package())


class _DevArray:
    """Minimal __cuda_array_interface__ carrier so that torch can alias a raw device pointer."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def device_tensor(ptr: int, n: int):
    import torch

    return torch.as_tensor(_DevArray(ptr, n), device="cuda")


class RingTransport:
    """Neighbour ring + all-gather over a torch.distributed process group."""

    def __init__(self, dist, rank: int, world: int, group=None):
        self.dist, self.rank, self.world = dist, rank, world
        self.group = group  # None: the default process group
        self.up = (rank + 1) % world
        self.down = (rank - 1) % world
        self.backend = dist.get_backend(group)
        self.host_staged = self.backend != "nccl"

    # -- ring -------------------------------------------------------------------------------
    def start_ring(self, send_dn, send_up, recv_lo, recv_hi):
        """send_up -> rank above's recv_lo, send_dn -> rank below's recv_hi.  Returns a handle
        for finish_ring().  Order of the P2P ops matters for world == 2, where both neighbours
        are the same peer: [send_up, send_dn] pairs with the peer's [recv_lo, recv_hi]."""
        dist = self.dist
        if self.host_staged:
            bufs = [send_up.cpu(), send_dn.cpu(), recv_lo.cpu(), recv_hi.cpu()]
        else:
            bufs = [send_up, send_dn, recv_lo, recv_hi]
        g = self.group
        ops = [
            dist.P2POp(dist.isend, bufs[0], self.up, group=g),
            dist.P2POp(dist.isend, bufs[1], self.down, group=g),
            dist.P2POp(dist.irecv, bufs[2], self.down, group=g),
            dist.P2POp(dist.irecv, bufs[3], self.up, group=g),
        ]
        return dist.batch_isend_irecv(ops), bufs, (recv_lo, recv_hi)

    def finish_ring(self, handle):
        reqs, bufs, (recv_lo, recv_hi) = handle
        for r in reqs:
            r.wait()
        if self.host_staged:
            recv_lo.copy_(bufs[2])
            recv_hi.copy_(bufs[3])

    def ring(self, send_dn, send_up, recv_lo, recv_hi):
        self.finish_ring(self.start_ring(send_dn, send_up, recv_lo, recv_hi))

    # -- all-gather (rank-major) --------------------------------------------------------------
    def allgather(self, local, gathered):
        dist = self.dist
        if self.host_staged:
            src = local.cpu()
            parts = [src.new_empty(src.shape) for _ in range(self.world)]
            dist.all_gather(parts, src, group=self.group)
            n = src.numel()
            for r, p in enumerate(parts):
                gathered[r * n : (r + 1) * n].copy_(p)
        else:
            dist.all_gather_into_tensor(gathered, local, group=self.group)


class _SlabBuffers:
    """torch views of one slab context's exchange buffers."""

    def __init__(self, sol: Solver):
        self.halo = [device_tensor(*sol.buffer("halo", k)) for k in range(4)]  # send_dn, send_up, recv_lo, recv_hi
        self.phi = [device_tensor(*sol.buffer("phi", k)) for k in range(4)]
        self.edge_local = device_tensor(*sol.buffer("edge", 0))
        self.edge_all = device_tensor(*sol.buffer("edge", 1))


class DistributedSlab:
    """One rank of a z-slab run.  Mirrors Solver's reference-named methods."""

    def __init__(self, params: Params, rank: int, world: int, dist, group=None):
        import torch

        self.torch = torch
        self.p = params.copy()
        self.rank, self.world = rank, world
        self.solver = Solver(params, rank, world, slab=True)
        self.stream = torch.cuda.Stream()
        self.solver.set_stream(self.stream.cuda_stream)
        self.buf = _SlabBuffers(self.solver)
        self.tr = RingTransport(dist, rank, world, group)

    def close(self):
        self.solver.close()

    # -- Poisson (fast_Poisson, poisson.cu:75-103, across slabs) -------------------------------
    def fast_Poisson(self):
        s, b, torch = self.solver, self.buf, self.torch
        with torch.cuda.stream(self.stream):
            s.call("poisson_stage1")
            self.tr.allgather(b.edge_local, b.edge_all)
            s.call("poisson_stage2")
            s.call("phi_halo_pack")
            self.tr.ring(b.phi[0], b.phi[1], b.phi[2], b.phi[3])
            s.call("poisson_stage3")

    # -- LBM (stream_collide_save, LBM.cu:465-481, with halo/compute overlap) ------------------
    def stream_collide_save(self, t: float = 0.0):
        s, b, torch = self.solver, self.buf, self.torch
        with torch.cuda.stream(self.stream):
            s.call("collide_boundary_planes")
            s.call("halo_pack")
            h = self.tr.start_ring(b.halo[0], b.halo[1], b.halo[2], b.halo[3])
            s.call("collide_interior_planes")  # overlaps the exchange
            self.tr.finish_ring(h)
            s.call("halo_unpack")

    def step(self, n: int = 1):
        for _ in range(n):  # main.cu:189-200
            self.stream_collide_save()
            self.fast_Poisson()
            self.solver.call("advance_time")

    # -- initial state --------------------------------------------------------------------------
    def initialization(self):
        """initialization(), LBM.cu:68-109, with the slab Poisson in the PB loop."""
        s = self.solver
        with self.torch.cuda.stream(self.stream):
            s.call("init_fields")
            s.call("pbe_begin")
        for _ in range(self.p.pb_iterations):
            with self.torch.cuda.stream(self.stream):
                s.call("pbe_concentrations")
            self.fast_Poisson()
            with self.torch.cuda.stream(self.stream):
                s.call("pbe_relax")
        s.call("pbe_end")

    def init_equilibrium(self):
        with self.torch.cuda.stream(self.stream):
            self.solver.init_equilibrium()

    def synchronize(self):
        self.solver.synchronize()

    # -- diagnostics (main.cu:211-222): every slab reduces its own planes, the ranks combine ----
    def _allreduce(self, value: float, op):
        t = self.torch.tensor([value], dtype=self.torch.float64, device="cpu" if self.tr.host_staged else "cuda")
        self.tr.dist.all_reduce(t, op=op, group=self.tr.group)
        return float(t.item())

    def current(self) -> float:
        """Wall current: only the slab holding the upper plate contributes (ekpnp_current)."""
        return self._allreduce(self.solver.current(), self.tr.dist.ReduceOp.SUM)

    def umax(self) -> float:
        return self._allreduce(self.solver.umax(), self.tr.dist.ReduceOp.MAX)


class LocalSlabGroup:
    """P slab contexts in one process on one device (tests): same kernels, same call order,
    device-to-device copies instead of RCCL."""

    def __init__(self, params: Params, nslabs: int):
        import torch

        self.torch = torch
        self.p = params.copy()
        self.n = nslabs
        self.stream = torch.cuda.Stream()
        self.sol = [Solver(params, r, nslabs, slab=True) for r in range(nslabs)]
        for s in self.sol:
            s.set_stream(self.stream.cuda_stream)
        self.buf = [_SlabBuffers(s) for s in self.sol]

    def close(self):
        for s in self.sol:
            s.close()

    def _each(self, name):
        for s in self.sol:
            s.call(name)

    def _ring(self, kind):
        n = self.n
        for r in range(n):
            b = getattr(self.buf[r], kind)
            up, dn = (r + 1) % n, (r - 1) % n
            getattr(self.buf[up], kind)[2].copy_(b[1])  # my send_up -> upper neighbour's recv_lo
            getattr(self.buf[dn], kind)[3].copy_(b[0])  # my send_dn -> lower neighbour's recv_hi

    def fast_Poisson(self):
        with self.torch.cuda.stream(self.stream):
            self._each("poisson_stage1")
            per = self.buf[0].edge_local.numel()
            for r in range(self.n):
                for q in range(self.n):
                    self.buf[q].edge_all[r * per : (r + 1) * per].copy_(self.buf[r].edge_local)
            self._each("poisson_stage2")
            self._each("phi_halo_pack")
            self._ring("phi")
            self._each("poisson_stage3")

    def stream_collide_save(self, t: float = 0.0):
        with self.torch.cuda.stream(self.stream):
            self._each("collide_boundary_planes")
            self._each("halo_pack")
            self._ring("halo")
            self._each("collide_interior_planes")
            self._each("halo_unpack")

    def step(self, n: int = 1):
        for _ in range(n):
            self.stream_collide_save()
            self.fast_Poisson()
            self._each("advance_time")

    def initialization(self):
        with self.torch.cuda.stream(self.stream):
            self._each("init_fields")
            self._each("pbe_begin")
        for _ in range(self.p.pb_iterations):
            with self.torch.cuda.stream(self.stream):
                self._each("pbe_concentrations")
            self.fast_Poisson()
            with self.torch.cuda.stream(self.stream):
                self._each("pbe_relax")
        self._each("pbe_end")

    def init_equilibrium(self):
        with self.torch.cuda.stream(self.stream):
            for s in self.sol:
                s.init_equilibrium()

    def synchronize(self):
        for s in self.sol:
            s.synchronize()

    def current(self) -> float:
        return sum(s.current() for s in self.sol)

    def umax(self) -> float:
        return max(s.umax() for s in self.sol)

    # -- whole-lattice views ------------------------------------------------------------------
    def fields(self) -> dict:
        return {k: np.concatenate([s.get_field(k) for s in self.sol], axis=0) for k in FIELDS}

    def set_fields(self, d: dict):
        for k, v in d.items():
            a = np.asarray(v, dtype=np.float64)
            for s in self.sol:
                s.set_field(k, a[s.z0 : s.z0 + s.nz_local])
