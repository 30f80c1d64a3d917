"""RCCL leg of RingTransport on ONE rank: the ring and the all-gather with world_size 1, i.e. both
neighbours are the rank itself (the same two-sends/two-receives-to-one-peer pattern as world 2),
on tensors that alias raw device pointers of a Solver, as DistributedSlab does.
    python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 tools/nccl_self_ring.py
"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
pkg = G.load_package()
from examples.host_transport import RingTransport, device_tensor

p = pkg.default_params(64, 64, 16)
with pkg.Solver(p) as s:
    n = 64 * 64 * 16
    names = ["rho", "ux", "uy", "uz", "c", "cn"]
    for i, k in enumerate(names):
        s.set_field(k, np.full(s.shape, float(i + 1)))
    t = {k: device_tensor(s.field_device_ptr(k), n) for k in names}
    tr = RingTransport(dist, 0, 1)
    assert not tr.host_staged
    stream = torch.cuda.Stream()
    s.set_stream(stream.cuda_stream)
    with torch.cuda.stream(stream):
        for rep in range(3):
            h = tr.start_ring(t["rho"], t["ux"], t["uy"], t["uz"])   # send_dn=rho, send_up=ux, recv_lo=uy, recv_hi=uz
            tr.finish_ring(h)
        tr.allgather(t["c"], t["cn"])
    stream.synchronize()
    uy, uz, cn = s.get_field("uy"), s.get_field("uz"), s.get_field("cn")
    assert np.all(uy == 2.0), uy.ravel()[:4]   # recv_lo <- send_up (ux = 2)
    assert np.all(uz == 1.0), uz.ravel()[:4]   # recv_hi <- send_dn (rho = 1)
    assert np.all(cn == 5.0)
print("RCCL self-ring ok: send_up -> recv_lo, send_dn -> recv_hi, all_gather_into_tensor", flush=True)
dist.destroy_process_group()
