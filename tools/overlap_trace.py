"""overlap_trace.py <trace_kernel_trace.csv> - where the halo-exchange kernel sits relative to the
collision of the interior planes in a rocprofv3 --kernel-trace of the multi-rank code path
(tools/profile_slab.sh).  For every step: start / end of the interior SWEEP (one k_collide_bulk launch
behind its short lead-in launch in two-buffer mode; a sequence of launches of up to 64 planes in in-place
mode - the sweep is then the span from its first to its last launch) and of every RCCL kernel that overlaps
it, relative to the sweep's start, in milliseconds."""
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
rccl = [r for r in rows if "rccl" in r["Kernel_Name"].lower() or "nccl" in r["Kernel_Name"].lower()]
# a step's sweep: the bulk launches between two consecutive halo unpack kernels
marks = [r["s"] for r in rows if "k_halo_unpack" in r["Kernel_Name"]]
bulk = [r for r in rows if "k_collide_bulk<" in r["Kernel_Name"] and "true>" in r["Kernel_Name"]]
out = []
prev = 0
for mk in marks:
    # interior launches of this step: everything bigger than a plane or two (the boundary planes are single-plane launches)
    mine = [b for b in bulk if prev <= b["s"] < mk]
    prev = mk
    if not mine:
        continue
    big = [b for b in mine if int(b["Grid_Size_X"]) >= max(int(x["Grid_Size_X"]) for x in mine) // 64]
    planes1 = min(int(b["Grid_Size_X"]) for b in mine)
    sweep = [b for b in mine if int(b["Grid_Size_X"]) > planes1] or big
    s0, e1 = min(b["s"] for b in sweep), max(b["e"] for b in sweep)
    rec = {"sweep_ms": round((e1 - s0) / 1e6, 3), "sweep_launches": len(sweep), "first_launch_ms": round((sweep[0]["e"] - sweep[0]["s"]) / 1e6, 3), "rccl": []}
    for c in rccl:
        if c["s"] < e1 and c["e"] > s0 - 2_000_000:
            rec["rccl"].append({"start_ms_after_sweep_start": round((c["s"] - s0) / 1e6, 3), "end_ms_after_sweep_start": round((c["e"] - s0) / 1e6, 3),
                                "duration_ms": round((c["e"] - c["s"]) / 1e6, 3), "end_ms_before_sweep_end": round((e1 - c["e"]) / 1e6, 3),
                                "grid": [int(c["Grid_Size_X"]), int(c["Workgroup_Size_X"])]})
    out.append(rec)
print(json.dumps({"steps": out, "rccl_kernels_total": len(rccl), "steps_seen": len(out)}, indent=1))
