"""trace_steps.py - cut a rocprofv3 --kernel-trace CSV of the slab path into time steps.

A time step of the slab path (csrc/slab_team.hip: team_stream_collide_save) starts with the launch(es) of
ekpnp_collide_boundary_planes - k_collide_faces (both faces in one launch, the default), or k_collide_wall /
k_collide_edge (one launch per face) - on every slab the process drives, then the halo exchange goes to the comm
stream and the interior sweep (k_collide_bulk launches) runs beside it.  So: a step begins at the first boundary-plane
kernel that follows a bulk launch (or that opens the trace), and lasts to the next such kernel.  Nothing here depends on
k_halo_pack / k_halo_unpack: they left the step in round 4 (the edge planes write / read the exchange buffers
themselves) and only come back under EKPNP_HALO_DIRECT=0.

Used by overlap_trace.py, step_timeline.py, gpu_busy_from_trace.py; tests/test_tools_cpu.py runs it on a committed
excerpt of a real trace (tests/golden/trace_excerpt_slab.csv) so that a renamed kernel shows up as a failing test,
not as an empty profile."""
import csv

BOUNDARY = ("k_collide_faces", "k_collide_wall", "k_collide_edge")
BULK = "k_collide_bulk<"
PACK = ("k_halo_pack", "k_halo_unpack")


def short_name(name):
    return name.split("(")[0].replace("void ", "").replace("ekpnp::", "")


def is_rccl(name):
    n = name.lower()
    return "rccl" in n or "nccl" in n


def load_rows(path):
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append({"s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"]), "n": r["Kernel_Name"],
                         "q": r.get("Stream_Id", r.get("Queue_Id", "?")), "gx": int(r.get("Grid_Size_X", 0) or 0),
                         "wx": int(r.get("Workgroup_Size_X", 0) or 0)})
    rows.sort(key=lambda r: r["s"])
    return rows


def step_starts(rows):
    """indices into rows (start order) of the first boundary-plane kernel of every time step"""
    starts, in_boundary = [], False
    for i, r in enumerate(rows):
        if any(b in r["n"] for b in BOUNDARY):
            if not in_boundary:
                starts.append(i)
            in_boundary = True
        elif BULK in r["n"]:
            in_boundary = False
    return starts


def split_steps(rows):
    """list of row lists, one per COMPLETE step (from one step start to the next; the trace's last, open step is
    closed at the last kernel that belongs to it: everything up to the end of the trace)"""
    st = step_starts(rows)
    return [rows[a:b] for a, b in zip(st, st[1:] + [len(rows)])]


def sweep_of(step):
    """the interior sweep of a step: the k_collide_bulk launches behind the boundary planes (and behind the pack kernel
    when EKPNP_HALO_DIRECT=0 launches the edge planes as one-plane bulk launches in front of it)"""
    last_pre = max((i for i, r in enumerate(step) if any(b in r["n"] for b in BOUNDARY + PACK[:1])), default=-1)
    return [r for r in step[last_pre + 1:] if BULK in r["n"]]


def write_excerpt(path, out_path, nsteps=3):
    """the rows of the last `nsteps` complete steps of a trace (plus the kernel that opens the following one), with the
    columns the tools read and timestamps rebased to the first row: a fixture for tests/test_tools_cpu.py"""
    rows = load_rows(path)
    st = step_starts(rows)
    if len(st) < nsteps + 1:
        raise SystemExit(f"trace_steps: {len(st)} step starts in {path}, {nsteps + 1} needed")
    part = rows[st[-nsteps - 1]:st[-1] + 1]
    t0 = part[0]["s"]
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Stream_Id", "Kernel_Name", "Start_Timestamp", "End_Timestamp", "Workgroup_Size_X", "Grid_Size_X"])
        for r in part:
            w.writerow([r["q"], short_name(r["n"]) if "ekpnp" in r["n"] else r["n"].split("(")[0], r["s"] - t0, r["e"] - t0, r["wx"], r["gx"]])


if __name__ == "__main__":
    import sys

    if len(sys.argv) >= 4 and sys.argv[1] == "--excerpt":
        write_excerpt(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 3)
    else:
        raise SystemExit("usage: trace_steps.py --excerpt <kernel_trace.csv> <out.csv> [nsteps]")
