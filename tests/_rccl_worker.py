"""Worker of tests/test_slab_gpu.py::test_native_rccl_ranks_sharing_one_gpu: one process per rank, the
library's OWN transport (ekpnp_slab_attach_comm: ncclCommInitRank, send/recv ring, all-gather,
all-reduce), all ranks on the one GPU of the test box.  RCCL refuses two ranks of one host on one
device, so every rank claims to be a different host (NCCL_HOSTID) and RCCL connects them through its
socket transport over loopback: the library's code path is exactly the multi-GPU one, only RCCL's
wire differs.  torch.distributed (gloo) is the control plane that hands the id round, as in bench.py."""
import os
import sys

import numpy as np

RANK = int(os.environ["RANK"])
os.environ["NCCL_HOSTID"] = f"ekpnp-test-rank{RANK}"  # before librccl is loaded
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
os.environ.setdefault("NCCL_IB_DISABLE", "1")

# ranks sharing one device must not race for its memory in the placement search of ekpnp_create
os.environ.setdefault("EKPNP_PLACEMENT_TRIES", "1")
# ... and must not oversubscribe its hardware queues (include/ekpnp.h: ekpnp_plane_transforms); read when the HIP runtime starts
os.environ.setdefault("GPU_MAX_HW_QUEUES", "1")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    pkg = G.load_package()
    out = os.environ["EKPNP_SLAB_OUT"]
    nx, ny, nz = (int(v) for v in os.environ["EKPNP_SLAB_GRID"].split("x"))
    p = pkg.default_params(nx, ny, nz)
    p.pb_iterations = 12
    p.in_place = int(os.environ.get("EKPNP_SLAB_IN_PLACE", "0"))
    s = pkg.Solver(p, rank, world, slab=True)
    ident = [pkg.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ident, src=0)
    s.attach_comm(ident[0])  # collective; from here every verb of the context is the whole-lattice one
    s.initialization()
    start = np.load(os.path.join(out, "start.npz"))
    for k in ("rho", "c", "cn", "T", "ux", "uy", "uz"):
        s.set_field(k, start[k][s.z0 : s.z0 + s.nz_local])
    s.fast_Poisson()
    s.init_equilibrium()
    midway = os.environ.get("EKPNP_RCCL_TUNE_MIDWAY", "")  # "knob=value,...": ekpnp_tune on the live transport after half of the steps
    if midway:
        s.step(3)
        for kv in midway.split(","):
            k, v = kv.split("=")
            s.tune(k, int(v))
        s.step(3)
    else:
        s.step(6)
    s.synchronize()
    current, umax = s.current(), s.umax()  # all-reduce inside the library
    fields = s.fields()
    if os.environ.get("EKPNP_RCCL_FIELDS_ONLY") == "1":  # full-width planes: fields and diagnostics only (the text files would be GBs)
        pt = s.plane_transforms()
        np.savez(os.path.join(out, f"rank{rank}.npz"), z0=s.z0, current=current, umax=umax, own_passes=pt["own_passes"], ranks_on_device=pt["ranks_on_device"], **fields)
        dist.barrier()
        s.close()
        dist.destroy_process_group()
        return
    # whole-lattice files, written plane by plane in rank order by the ranks taking turns
    s.save_data_end(os.path.join(out, "data_end.dat"), 0.25)
    s.save_data_tecplot(os.path.join(out, "tec.dat"), 0.25)
    ckpt = os.path.join(out, f"ckpt_rank{rank}.bin")  # per-rank, self-contained (ghost planes travel)
    s.save_checkpoint(ckpt)
    # bitwise continuation from the checkpoint, against simply continuing
    s.step(3)
    cont = s.fields()
    t_ck = s.load_checkpoint(ckpt)
    s.step(3)
    again = s.fields()
    same = all(np.array_equal(cont[k], again[k]) for k in cont)
    # a file that cannot be opened: EVERY rank gets the error (the ranks agree on the status between the turns),
    # nobody hangs, and the transport keeps working afterwards
    try:
        s.save_data_end(os.path.join(out, "no_such_directory", "x.dat"), 0.25)
        io_error = ""
    except pkg.EkpnpError as e:
        io_error = str(e)
    umax_after = s.umax()
    # the reference's restart route (fields -> equilibrium), collective read of the whole-lattice file
    t_read = s.read_data(os.path.join(out, "data_end.dat"))
    reread = s.fields()
    np.savez(os.path.join(out, f"rank{rank}.npz"), z0=s.z0, current=current, umax=umax, ckpt_same=same, t_ck=t_ck, t_read=t_read, io_error=io_error, umax_after=umax_after,
             **fields, **{"re_" + k: v for k, v in reread.items()})
    dist.barrier()
    s.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
