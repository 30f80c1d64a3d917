"""overlap_trace.py <trace_kernel_trace.csv> - where the halo-exchange kernel sits relative to the
collision of the interior planes in a rocprofv3 --kernel-trace of the multi-rank code path
(tools/profile_slab.sh).  For every step: start / end of k_collide_bulk (interior planes) and of
every RCCL kernel that overlaps it, relative to the bulk kernel's start, in milliseconds."""
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
bulk = [r for r in rows if "k_collide_bulk<4, true>" in r["Kernel_Name"] and int(r["Grid_Size_X"]) > 100000 * 64]
rccl = [r for r in rows if "rccl" in r["Kernel_Name"].lower() or "nccl" in r["Kernel_Name"].lower()]
out = []
for b in bulk:
    inside = [c for c in rccl if c["s"] < b["e"] and c["e"] > b["s"]]
    rec = {"bulk_ms": round((b["e"] - b["s"]) / 1e6, 3), "bulk_stream": b["Stream_Id"], "rccl": []}
    for c in inside:
        rec["rccl"].append({"stream": c["Stream_Id"], "start_ms_after_bulk_start": round((c["s"] - b["s"]) / 1e6, 3),
                            "end_ms_after_bulk_start": round((c["e"] - b["s"]) / 1e6, 3), "end_ms_before_bulk_end": round((b["e"] - c["e"]) / 1e6, 3),
                            "grid": [int(c["Grid_Size_X"]), int(c["Workgroup_Size_X"])]})
    out.append(rec)
print(json.dumps({"steps": out, "rccl_kernels_total": len(rccl), "bulk_launches": len(bulk)}, indent=1))
