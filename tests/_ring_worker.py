"""Worker of tests/test_slab_cpu.py: run under torch.distributed.run with the gloo backend.
Exercises examples/host_transport.py's RingTransport (the host-side transport example) on CPU
tensors: ring semantics (send_up -> upper neighbour's recv_lo, send_dn -> lower neighbour's
recv_hi, wrap-around) and the rank-major all-gather."""
import os
import sys

# ranks sharing one device must not race for its memory in the placement search of ekpnp_create
os.environ.setdefault("EKPNP_PLACEMENT_TRIES", "1")
# ... and must not oversubscribe its hardware queues (include/ekpnp.h: ekpnp_plane_transforms); read when the HIP runtime starts
os.environ.setdefault("GPU_MAX_HW_QUEUES", "1")

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    G.load_package()
    from examples.host_transport import RingTransport, slab_extent

    tr = RingTransport(dist, rank, world)
    assert tr.host_staged
    n = 1000
    for rep in range(3):  # repeated exchanges must not cross-talk
        send_dn = torch.full((n,), 10.0 * rank + 1 + 100 * rep, dtype=torch.float64)
        send_up = torch.full((n,), 10.0 * rank + 2 + 100 * rep, dtype=torch.float64)
        recv_lo = torch.zeros(n, dtype=torch.float64)
        recv_hi = torch.zeros(n, dtype=torch.float64)
        h = tr.start_ring(send_dn, send_up, recv_lo, recv_hi)
        tr.finish_ring(h)
        below, above = (rank - 1) % world, (rank + 1) % world
        assert torch.all(recv_lo == 10.0 * below + 2 + 100 * rep), (rank, recv_lo[0])  # the rank below sent UP
        assert torch.all(recv_hi == 10.0 * above + 1 + 100 * rep), (rank, recv_hi[0])  # the rank above sent DOWN
    local = torch.arange(8, dtype=torch.float64) + 100 * rank
    gathered = torch.zeros(8 * world, dtype=torch.float64)
    tr.allgather(local, gathered)
    for r in range(world):
        assert torch.all(gathered[8 * r : 8 * r + 8] == torch.arange(8, dtype=torch.float64) + 100 * r)
    z0, nzl = slab_extent(64 * world, rank, world)
    assert (z0, nzl) == (rank * 64, 64)
    dist.barrier()
    if rank == 0:
        open(os.environ["EKPNP_RING_OK"], "w").write(f"ok {world}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
