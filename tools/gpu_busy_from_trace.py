"""gpu_busy_from_trace.py <kernel_trace.csv> [last_fraction] - how much of the wall time of a run was the GPU idle?

Union of all kernel intervals (device copies show up as __amd_rocclr_copyBuffer kernels) of a rocprofv3 --kernel-trace,
over the last `last_fraction` (default 0.5: the timed steps, past set-up and warm-up) of the traced span.  If one host
thread driving 8 slabs could not keep the device fed, it would show here as idle gaps between kernels."""
import csv
import json
import sys

rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t1 - int((t1 - t0) * frac)
iv = [(max(s, lo), e) for s, e, _ in rows if e > lo]
busy, cur_s, cur_e, gaps = 0, None, None, []
for s, e in sorted(iv):
    if cur_s is None:
        cur_s, cur_e = s, e
    elif s <= cur_e:
        cur_e = max(cur_e, e)
    else:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
busy += cur_e - cur_s
span = t1 - lo
gaps.sort(reverse=True)
print(json.dumps({"window_ms": round(span / 1e6, 3), "busy_ms": round(busy / 1e6, 3), "idle_ms": round((span - busy) / 1e6, 3), "idle_fraction": round((span - busy) / span, 5),
                  "gaps": len(gaps), "largest_gaps_us": [round(g / 1e3, 1) for g in gaps[:5]], "kernels_in_window": len(iv)}))
