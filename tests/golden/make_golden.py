"""Generates tests/golden/ref_*.npz from the REFERENCE's own kernels.  Run on the GPU box:

    python tests/golden/make_golden.py gpurun_out/golden [NXxNYxNZ ...]

Without a grid: the reference's own compile-time grid 50x8x51, cases G1-G7 below.  With grids:
the second build(s) of the reference's kernels that oracle/build_ref.sh makes for other grids
(sed on the temporary copy of LBM.h:32-35,40-42 only) -> ref_<grid>_g{1,2,4,5}_full.npz.  These
pin the oracle on more than one size and cover what 50x8x51 cannot: rows of more than one
64-node tile (130 = three tiles, the x+-1 pull across a tile boundary) and channels taller than
the 66 planes the cyclic-reduction z solve handles (83 planes: the serial Thomas sweep).

It drives oracle/_ref/ref_driver (built by oracle/build_ref.sh from /root/reference: the
reference's LBM.cu / poisson.cu kernels compiled for gfx950) on the reference's compile-time
grid 50x8x51 (LBM.h:32-35) and stores the macroscopic fields as FP64:

  ref_g1.npz  G1: state after initialization() (501 PB sweeps) and after 1/5/20/100 steps
              of the default, x-y uniform run (z profiles: the fields are uniform in x,y)
  ref_g2.npz  G2: the closed-form 3-D perturbation of SURVEY.md §8(c) on top of G1's initial
              state -> fast_Poisson -> init_equilibrium -> 1, 2, 50 steps (full 3-D fields)
  ref_g5.npz  G5: fast_Poisson alone on random c, cn

Only data is stored (inputs and the reference's outputs); no reference source travels.
The files written under the output directory are copied into tests/golden/ by hand
(see DESIGN.md "Oracle").
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

NX, NY, NZ = 50, 8, 51
SHAPE = (NZ, NY, NX)
N = NX * NY * NZ


def read_bin(path):
    a = np.fromfile(path, dtype=np.float64)
    assert a.size == 11 * N, (path, a.size)
    return {k: a[i * N : (i + 1) * N].reshape(SHAPE).copy() for i, k in enumerate(O.FIELDS)}


def read_current(path):
    return float(np.fromfile(path + ".current", dtype=np.float64)[0])


def write_bin(path, f):
    np.concatenate([np.ascontiguousarray(f[k], dtype=np.float64).ravel() for k in O.FIELDS]).tofile(path)


def io_fields(seed=6):
    """Seeded fields with realistic magnitudes for the IO / diagnostics golden (G6)."""
    rng = np.random.default_rng(seed)
    f = {}
    f["rho"] = 1000.0 + 1e-3 * rng.standard_normal(SHAPE)
    f["c"] = 0.0105 + 1e-3 * rng.random(SHAPE)
    f["cn"] = 0.0095 + 1e-3 * rng.random(SHAPE)
    f["phi"] = -5e-3 * rng.random(SHAPE)
    f["ux"] = 1e-4 * rng.standard_normal(SHAPE)
    f["uy"] = 1e-4 * rng.standard_normal(SHAPE)
    f["uz"] = 1e-4 * rng.standard_normal(SHAPE)
    f["Ex"] = 1e2 * rng.standard_normal(SHAPE)
    f["Ey"] = 1e2 * rng.standard_normal(SHAPE)
    f["Ez"] = 5e4 * rng.standard_normal(SHAPE)
    f["T"] = rng.random(SHAPE)
    return f


def set_grid(nx, ny, nz):
    global NX, NY, NZ, SHAPE, N
    NX, NY, NZ = nx, ny, nz
    SHAPE = (NZ, NY, NX)
    N = NX * NY * NZ


def extra_grid(outdir, grid):
    """G1 (init + 1/5/20 steps), G2 (0/1/2/50), G4 and G5 on another compile-time grid."""
    nx, ny, nz = (int(v) for v in grid.split("x"))
    set_grid(nx, ny, nz)
    os.makedirs(outdir, exist_ok=True)
    drv = os.path.join(ROOT, "oracle", "_ref", "ref_driver_" + grid)
    tmp = tempfile.mkdtemp()

    def run(*args):
        print("+ ref_driver_" + grid, *args, flush=True)
        subprocess.check_call([drv, tmp, *args])

    def trace(path, n):
        return np.fromfile(path, dtype=np.float64).reshape(n, 2, NZ)

    p = O.default_params(NX, NY, NZ)  # Lx = NX dx, Ly = NY dy, Lz = (NZ-1) dz: what build_ref.sh wrote
    marks1 = (1, 5, 20)
    run("init", "g1", *[str(m) for m in marks1])
    g1 = {"marks": np.array(marks1), "grid": np.array([NX, NY, NZ])}
    init = read_bin(os.path.join(tmp, "g1_init.bin"))
    for tag, f in [("init", init)] + [(f"step{m}", read_bin(os.path.join(tmp, f"g1_step{m}.bin"))) for m in marks1]:
        for k, v in f.items():
            g1[f"{tag}_{k}"] = v
    g1["current"] = np.array([read_current(os.path.join(tmp, f"g1_step{m}.bin")) for m in marks1])
    g1["init_trace"] = trace(os.path.join(tmp, "g1_init_trace.bin"), 501)
    g1["step_trace"] = trace(os.path.join(tmp, "g1_step_trace.bin"), marks1[-1])
    np.savez_compressed(os.path.join(outdir, f"ref_{grid}_g1_full.npz"), **g1)

    start = O.perturb_fields(p, init)
    inp = os.path.join(tmp, "g2_in.bin")
    write_bin(inp, start)
    run("fields", inp, "g2", "1", "2", "50")
    g2 = {"marks": np.array([1, 2, 50]), "grid": np.array([NX, NY, NZ])}
    for k, v in start.items():
        g2["input_" + k] = v
    for m in (0, 1, 2, 50):
        f = read_bin(os.path.join(tmp, f"g2_step{m}.bin"))
        for k, v in f.items():
            g2[f"step{m}_{k}"] = v
    g2["current"] = np.array([read_current(os.path.join(tmp, f"g2_step{m}.bin")) for m in (1, 2, 50)])
    g2["step_trace"] = trace(os.path.join(tmp, "g2_step_trace.bin"), 51)
    np.savez_compressed(os.path.join(outdir, f"ref_{grid}_g2_full.npz"), **g2)

    run("kernels", inp, "g4")
    npop = 4 * 27 * NZ * 3 * NX
    g4 = {}
    for stage in ("step1", "collide", "boundary", "stream", "bc_charge"):
        a = np.fromfile(os.path.join(tmp, f"g4_{stage}.bin"), dtype=np.float64)
        assert a.size == npop, (stage, a.size)
        g4[stage] = a.reshape(4, 27, NZ, 3, NX)
    f0 = read_bin(os.path.join(tmp, "g4_fields0.bin"))
    for k in O.FIELDS:
        g4["fields0_" + k] = f0[k]
    np.savez_compressed(os.path.join(outdir, f"ref_{grid}_g4_full.npz"), **g4)

    rng = np.random.default_rng(5)
    f5 = {k: np.zeros(SHAPE) for k in O.FIELDS}
    f5["c"] = 0.01 * (1 + 0.2 * rng.random(SHAPE))
    f5["cn"] = 0.01 * (1 + 0.2 * rng.random(SHAPE))
    inp5 = os.path.join(tmp, "g5_in.bin")
    write_bin(inp5, f5)
    run("poisson", inp5, "g5")
    out5 = read_bin(os.path.join(tmp, "g5.bin"))
    g5 = {"grid": np.array([NX, NY, NZ])}
    for k in ("c", "cn"):
        g5["input_" + k] = f5[k]
    for k in ("phi", "Ex", "Ey", "Ez"):
        g5["out_" + k] = out5[k]
    np.savez_compressed(os.path.join(outdir, f"ref_{grid}_g5_full.npz"), **g5)
    print("golden vectors of grid", grid, "written to", outdir)


# G8: every physics knob away from its default and the two plates at DIFFERENT zeta potentials (all other
# goldens have voltage == voltage2, so a swapped plate could not be seen); written into the reference's
# __constant__/__device__ symbols at run time (ref_driver --set), LBM.h itself is not edited
ASYM = {"voltage": -3.1e-3, "voltage2": -6.9e-3, "Ext": 2.3e4, "TH": 0.7, "Ra": 1.7, "K": 3.9e-7, "Kn": -4.6e-7, "diffu": 1.2e-8,
        "diffun": 0.8e-8, "nu": 0.7e-6, "D": 1.1e-6, "exf": 2.0e7, "uw": 5.0e-4, "VC": 2.0e-6, "VCn": 0.5e-6, "V": 0.1, "VT": 0.07,
        "chargeinf": 0.012, "eps": 7.5e-10}


def asym_case(outdir, grid):
    """G8 on the default grid ("50x8x51") or an extra one: initialization() + 1/5/20 steps of the x-y uniform
    run and the perturbed 3-D run 0/1/30 steps, all with the ASYM physics."""
    nx, ny, nz = (int(v) for v in grid.split("x"))
    set_grid(nx, ny, nz)
    os.makedirs(outdir, exist_ok=True)
    drv = os.path.join(ROOT, "oracle", "_ref", "ref_driver" + ("" if grid == "50x8x51" else "_" + grid))
    tmp = tempfile.mkdtemp()
    sets = []
    for k, v in ASYM.items():
        sets += ["--set", f"{k}={v!r}"]

    def run(*args):
        print("+", os.path.basename(drv), *sets, *args, flush=True)
        subprocess.check_call([drv, tmp, *sets, *args])

    def trace(path, n):
        return np.fromfile(path, dtype=np.float64).reshape(n, 2, NZ)

    p = O.default_params(NX, NY, NZ)
    if grid == "50x8x51":
        p.Lx, p.Ly, p.Lz = 0.5e-6, 0.08e-6, 0.5e-6
    for k, v in ASYM.items():
        setattr(p, k, v)
    marks1 = (1, 5, 20)
    run("init", "a1", *[str(m) for m in marks1])
    g = {"grid": np.array([NX, NY, NZ]), "param_names": np.array(list(ASYM)), "param_values": np.array(list(ASYM.values())),
         "a1_marks": np.array(marks1)}
    init = read_bin(os.path.join(tmp, "a1_init.bin"))
    for tag, f in [("init", init)] + [(f"step{m}", read_bin(os.path.join(tmp, f"a1_step{m}.bin"))) for m in marks1]:
        for k, v in f.items():
            g[f"a1_{tag}_{k}"] = v
    g["a1_current"] = np.array([read_current(os.path.join(tmp, f"a1_step{m}.bin")) for m in marks1])
    g["a1_init_trace"] = trace(os.path.join(tmp, "a1_init_trace.bin"), 501)
    g["a1_step_trace"] = trace(os.path.join(tmp, "a1_step_trace.bin"), marks1[-1])
    start = O.perturb_fields(p, init)
    inp = os.path.join(tmp, "a2_in.bin")
    write_bin(inp, start)
    marks2 = (1, 30)
    run("fields", inp, "a2", *[str(m) for m in marks2])
    g["a2_marks"] = np.array(marks2)
    for k, v in start.items():
        g["a2_input_" + k] = v
    for m in (0,) + marks2:
        f = read_bin(os.path.join(tmp, f"a2_step{m}.bin"))
        for k, v in f.items():
            g[f"a2_step{m}_{k}"] = v
    g["a2_current"] = np.array([read_current(os.path.join(tmp, f"a2_step{m}.bin")) for m in marks2])
    g["a2_step_trace"] = trace(os.path.join(tmp, "a2_step_trace.bin"), 1 + marks2[-1])
    np.savez_compressed(os.path.join(outdir, f"ref_{grid}_g8_full.npz"), **g)
    print("asymmetric-physics vectors of grid", grid, "written to", outdir)


def main(outdir):
    set_grid(50, 8, 51)
    os.makedirs(outdir, exist_ok=True)
    drv = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    tmp = tempfile.mkdtemp()

    def run(*args):
        print("+ ref_driver", *args, flush=True)
        subprocess.check_call([drv, tmp, *args])

    p = O.default_params(NX, NY, NZ)
    p.Lx, p.Ly, p.Lz = 0.5e-6, 0.08e-6, 0.5e-6

    def trace(path, n):
        a = np.fromfile(path, dtype=np.float64)
        return a.reshape(n, 2, NZ)

    # ---- G1: default run, x-y uniform
    run("init", "g1", "1", "5", "20", "100")
    g1 = {"marks": np.array([1, 5, 20, 100])}
    init = read_bin(os.path.join(tmp, "g1_init.bin"))
    for tag, f in [("init", init)] + [(f"step{m}", read_bin(os.path.join(tmp, f"g1_step{m}.bin"))) for m in (1, 5, 20, 100)]:
        for k, v in f.items():
            g1[f"{tag}_{k}"] = v
    g1["current"] = np.array([read_current(os.path.join(tmp, f"g1_step{m}.bin")) for m in (1, 5, 20, 100)])
    g1["init_trace"] = trace(os.path.join(tmp, "g1_init_trace.bin"), 501)
    g1["step_trace"] = trace(os.path.join(tmp, "g1_step_trace.bin"), 100)
    np.savez_compressed(os.path.join(outdir, "ref_g1_full.npz"), **g1)

    # ---- G2: perturbed 3-D run
    start = O.perturb_fields(p, init)
    inp = os.path.join(tmp, "g2_in.bin")
    write_bin(inp, start)
    run("fields", inp, "g2", "1", "2", "50")
    g2 = {"marks": np.array([1, 2, 50])}
    for k, v in start.items():
        g2["input_" + k] = v
    for m in (0, 1, 2, 50):
        f = read_bin(os.path.join(tmp, f"g2_step{m}.bin"))
        for k, v in f.items():
            g2[f"step{m}_{k}"] = v
    g2["current"] = np.array([read_current(os.path.join(tmp, f"g2_step{m}.bin")) for m in (1, 2, 50)])
    g2["step_trace"] = trace(os.path.join(tmp, "g2_step_trace.bin"), 51)
    np.savez_compressed(os.path.join(outdir, "ref_g2_full.npz"), **g2)

    # ---- G4: per-kernel vectors on the G2 input (populations on the y rows {0,3,5})
    run("kernels", inp, "g4")
    npop = 4 * 27 * NZ * 3 * NX
    g4 = {}
    for stage in ("step1", "collide", "boundary", "stream", "bc_charge"):
        a = np.fromfile(os.path.join(tmp, f"g4_{stage}.bin"), dtype=np.float64)
        assert a.size == npop, (stage, a.size)
        g4[stage] = a.reshape(4, 27, NZ, 3, NX)
    f0 = read_bin(os.path.join(tmp, "g4_fields0.bin"))
    fc = read_bin(os.path.join(tmp, "g4_fields_collide.bin"))
    for k in O.FIELDS:
        g4["fields0_" + k] = f0[k]
        g4["fields_collide_" + k] = fc[k]
    np.savez_compressed(os.path.join(outdir, "ref_g4_full.npz"), **g4)

    # ---- G3: body-force driven channel (Poiseuille), no ions, no buoyancy: rho and u do not
    # depend on phi at all here (force = F (c - cn) E = 0), so they are free of the DC leak.
    run("--set", "exf=1e9", "--set", "chargeinf=0", "--set", "Ra=0", "--set", "TH=0", "init", "g3", "1", "100", "3000")
    g3 = {"marks": np.array([1, 100, 3000])}
    for m in (1, 100, 3000):
        f = read_bin(os.path.join(tmp, f"g3_step{m}.bin"))
        for k in ("rho", "ux", "uy", "uz", "c", "cn", "T"):
            g3[f"step{m}_{k}"] = f[k]
    np.savez_compressed(os.path.join(outdir, "ref_g3_full.npz"), **g3)

    # ---- G7: moving upper wall (uw = 1e-3, no ions, no buoyancy): the reference's wall-velocity terms
    # including the "+multis on direction 3 only" of LBM.cu:1904; rho and u do not see phi here.
    run("--set", "uw=1e-3", "--set", "chargeinf=0", "--set", "Ra=0", "--set", "TH=0", "init", "g7", "1", "100", "2000")
    g7 = {"marks": np.array([1, 100, 2000])}
    for m in (1, 100, 2000):
        f = read_bin(os.path.join(tmp, f"g7_step{m}.bin"))
        for k in ("rho", "ux", "uy", "uz"):
            g7[f"step{m}_{k}"] = f[k]
    np.savez_compressed(os.path.join(outdir, "ref_g7_full.npz"), **g7)

    # ---- G5
    rng = np.random.default_rng(5)
    f5 = {k: np.zeros(SHAPE) for k in O.FIELDS}
    f5["c"] = 0.01 * (1 + 0.2 * rng.random(SHAPE))
    f5["cn"] = 0.01 * (1 + 0.2 * rng.random(SHAPE))
    inp5 = os.path.join(tmp, "g5_in.bin")
    write_bin(inp5, f5)
    run("poisson", inp5, "g5")
    out5 = read_bin(os.path.join(tmp, "g5.bin"))
    g5 = {}
    for k, v in f5.items():
        g5["input_" + k] = v
    for k in ("phi", "Ex", "Ey", "Ez"):
        g5["out_" + k] = out5[k]
    np.savez_compressed(os.path.join(outdir, "ref_g5_full.npz"), **g5)
    # ---- G6: the reference's writers and diagnostics on seeded random fields
    f6 = io_fields()
    inp6 = os.path.join(tmp, "g6_in.bin")
    write_bin(inp6, f6)
    run("io", inp6, "g6")
    import hashlib
    import shutil

    g6 = {"input_sha256": np.array(hashlib.sha256(open(inp6, "rb").read()).hexdigest())}
    for name in ("data.dat", "data_end.dat", "umax.dat"):
        raw = open(os.path.join(tmp, "g6_" + name), "rb").read()
        g6[name + "_sha256"] = np.array(hashlib.sha256(raw).hexdigest())
        g6[name + "_size"] = np.array(len(raw))
        lines = raw.split(b"\n")
        g6[name + "_head"] = np.array(b"\n".join(lines[:6]).decode())
        g6[name + "_tail"] = np.array(b"\n".join(lines[-3:]).decode())
    g6["current"] = np.fromfile(os.path.join(tmp, "g6_current.bin"), dtype=np.float64)[0]
    np.savez_compressed(os.path.join(outdir, "ref_g6.npz"), **g6)
    shutil.copy(os.path.join(tmp, "g6_data_end.dat"), os.path.join(outdir, "g6_data_end.dat"))
    print("golden vectors written to", outdir)


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/golden"
    if len(sys.argv) > 2 and sys.argv[2] == "asym":
        for g in sys.argv[3:]:
            asym_case(out, g)
    elif len(sys.argv) > 2:
        for g in sys.argv[2:]:
            extra_grid(out, g)
    else:
        main(out)
