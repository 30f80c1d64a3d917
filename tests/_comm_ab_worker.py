"""Worker of tests/test_bench_cpu.py::test_comm_ab_legs_agree_over_two_gloo_ranks: bench.py's after-the-fact knob A/B
(comm_ab_leg) on two real gloo ranks with a stand-in context - what the ranks must AGREE on without a GPU: a knob that ONE
rank cannot set costs only its own leg on BOTH ranks (nobody steps into a collective alone), the timings are maxima over the
ranks, and every knob is back at its baseline afterwards."""
import importlib.util
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    calls = []

    class Fake:
        def tune(self, k, v):
            calls.append(["tune", k, v])
            if rank == 1 and k == "comm_cus" and v == 8:
                raise RuntimeError("rank 1: hipExtStreamCreateWithCUMask refused")

        def step(self, n):
            calls.append(["step", n])

        def kernel_timing(self, on):
            pass

        def kernel_timing_get(self):
            return 10, 100.0 * (rank + 1), 1

        def poisson_stage_timing_get(self):
            return 10, {"stage1": 1.0, "edge_exchange": 2.0 * (rank + 1), "stage2": 3.0, "phi_exchange": 0.5, "stage3": 0.1}

        def phase_timing_get(self):
            return 10, 30.0

        def comm_timing_get(self):
            return {k: {"n": 10, "wait_ms": 1.0 + rank, "transfer_ms": 1.0, "bytes_sent": 1} for k in ("halo", "edge", "phi")}

    f = Fake()
    base = b.ab_baseline()
    legs = [b.comm_ab_leg(label, knobs, f, f, 10, dist.barrier, dist, torch, world, base) for label, knobs in b.COMM_AB_LEGS]
    out = {"rank": rank, "legs": legs, "calls": calls}
    allout = [None] * world
    dist.all_gather_object(allout, out)
    if rank == 0:
        json.dump(allout, open(os.environ["EKPNP_AB_OUT"], "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
