"""copy_placement.py [n] [GiB] - does the rate of a PLAIN COPY depend on where its buffers lie in HBM?
n measurements of ekpnp_copy_bandwidth (two fresh buffers of GiB each, freed afterwards), a dummy allocation of
5.3 GiB added between measurements so that every pair of buffers lands in another physical region."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as G

pkg = G.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
gib = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
hold, vals = [], []
with pkg.Solver(pkg.default_params(16, 16, 17)) as s:
    for k in range(n):
        v = s.copy_bandwidth(int(gib * 2**30))
        vals.append(v)
        print(f"placement {k}: {v:8.1f} GB/s", flush=True)
        hold.append(torch.empty(int(5.3 * 2**30), dtype=torch.uint8, device="cuda"))
print(f"copy of 2 x {gib} GiB over {n} placements: min {min(vals):.1f} max {max(vals):.1f} GB/s, spread {(max(vals) / min(vals) - 1) * 100:.1f} %")
