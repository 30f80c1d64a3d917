"""copy_direction.py [GiB] [n] - is a device copy X -> Y as fast as Y -> X?  n pairs of buffers (each pair allocated after a
5.3 GiB dummy so that it lands elsewhere), both directions timed with HIP events (torch).  The interior sweep of a thin
lattice runs up to 8 % faster in one direction (A -> B) than in the other on the same placement
(profiles/r03_placement_search_after.log); this asks whether a plain copy shows it too."""
import sys
import torch

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
hold = []
def timed(dst, src):
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); dst.copy_(src); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return 2 * src.numel() / best / 1e6  # GB/s, read + write
for k in range(n):
    x = torch.zeros(int(gib * 2**30), dtype=torch.uint8, device="cuda")
    y = torch.zeros(int(gib * 2**30), dtype=torch.uint8, device="cuda")
    a, b = timed(y, x), timed(x, y)
    print(f"pair {k}: X->Y {a:8.1f} GB/s   Y->X {b:8.1f} GB/s   ratio {a / b:.4f}   X @ {x.data_ptr():#x}", flush=True)
    del x, y
    hold.append(torch.empty(int(5.3 * 2**30), dtype=torch.uint8, device="cuda"))
