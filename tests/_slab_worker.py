"""Worker of tests/test_slab_gpu.py: the true multi-process slab path (one process per rank, all
on the one GPU of the test box, gloo transport staged through the host)."""
import os
import sys

import numpy as np

if os.environ.get("EKPNP_SLAB_BACKEND", "gloo") == "nccl" and int(os.environ.get("WORLD_SIZE", "1")) > 1:
    # several RCCL ranks on ONE device: each claims its own host, RCCL wires them over its socket transport
    os.environ["NCCL_HOSTID"] = "ekpnp-test-rank" + os.environ["RANK"]
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    os.environ.setdefault("NCCL_IB_DISABLE", "1")

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
    torch.cuda.set_device(0)
    if os.environ.get("EKPNP_SLAB_BACKEND", "gloo") == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))  # RCCL
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    pkg = G.load_package()
    from examples.host_transport import DistributedSlab  # the host-side transport example (not product code)

    out = os.environ["EKPNP_SLAB_OUT"]
    nx, ny, nz = (int(v) for v in os.environ["EKPNP_SLAB_GRID"].split("x"))
    p = pkg.default_params(nx, ny, nz)
    p.pb_iterations = 12
    p.in_place = int(os.environ.get("EKPNP_SLAB_IN_PLACE", "0"))
    run = DistributedSlab(p, rank, world, dist)
    run.initialization()
    s = run.solver
    start = np.load(os.path.join(out, "start.npz"))
    for k in ("rho", "c", "cn", "T", "ux", "uy", "uz"):
        s.set_field(k, start[k][s.z0 : s.z0 + s.nz_local])
    run.fast_Poisson()
    run.init_equilibrium()
    run.step(6)
    run.synchronize()
    np.savez(os.path.join(out, f"rank{rank}.npz"), z0=s.z0, current=run.current(), umax=run.umax(), **s.fields())
    dist.barrier()
    run.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
