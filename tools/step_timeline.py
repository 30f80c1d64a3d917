"""step_timeline.py <kernel_trace.csv> [marker] - the kernels of the LAST full step of a traced run, in start order:
offset from the step's first kernel, duration, stream, and the gap since the previous kernel ended on the same stream.
A step runs from one kernel whose name contains `marker` to the next (default k_halo_pack: once per step on the
slab path).  Shows where a step's time goes between its big kernels."""
import csv
import json
import sys

rows = [dict(s=int(r["Start_Timestamp"]), e=int(r["End_Timestamp"]), n=r["Kernel_Name"], q=r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in csv.DictReader(open(sys.argv[1]))]
marker = sys.argv[2] if len(sys.argv) > 2 else "k_halo_pack"
rows.sort(key=lambda r: r["s"])
starts = [i for i, r in enumerate(rows) if marker in r["n"]]
a, b = starts[-2], starts[-1]
step = rows[a:b]
t0 = step[0]["s"]
last_end = {}
out = []
for r in step:
    gap = (r["s"] - last_end[r["q"]]) / 1e3 if r["q"] in last_end else None
    last_end[r["q"]] = max(last_end.get(r["q"], 0), r["e"])
    name = r["n"].split("(")[0].replace("void ", "").replace("ekpnp::", "")[:60]
    out.append({"at_ms": round((r["s"] - t0) / 1e6, 4), "ms": round((r["e"] - r["s"]) / 1e6, 4), "stream": r["q"], "gap_us": None if gap is None else round(gap, 1), "kernel": name})
print(json.dumps({"step_ms": round((rows[b]["s"] - t0) / 1e6, 4), "kernels": out}, indent=0))
