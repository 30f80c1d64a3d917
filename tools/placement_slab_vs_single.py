"""Does a SLAB context of a thin lattice land on slower HBM placements than a single context of the same lattice?
(round 4: profiles/r03_bench_thin_slabs_at_head.jsonl shows candidates of 10.2 - 10.9 ms per sweep for a 512x512x128 slab
where single contexts of the same lattice had 9.55 - 10.45).  Alternately creates both kinds in ONE process and prints what
the placement search of ekpnp_create timed (ekpnp_placement_report): both sweep directions averaged, every candidate.
    python tools/placement_slab_vs_single.py [NXxNYxNZ] [repeats] """
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G

pkg = G.load_package()
nx, ny, nz = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "512x512x128").split("x"))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
p = pkg.default_params(nx, ny, nz)
for r in range(reps):
    for kind in ("single", "slab"):
        s = pkg.Solver(p) if kind == "single" else pkg.Solver(p, 0, 1, slab=True)
        rep = s.placement_report()
        print(f"{kind:6s} {r}: tried {rep['tried']} chosen {rep['chosen']} sweep_ms {[round(v, 3) for v in rep['sweep_ms']]}", flush=True)
        s.close()
