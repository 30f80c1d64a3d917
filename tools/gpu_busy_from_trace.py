"""gpu_busy_from_trace.py <kernel_trace.csv> <steps> <marker_kernel> <markers_per_step> - how much of the wall time of the
last <steps> time steps of a run was the GPU idle?

Union of all kernel intervals (device copies show up as __amd_rocclr_copyBuffer kernels) of a rocprofv3 --kernel-trace.
The window is the last <steps> steps: it starts when the marker kernel (one that runs <markers_per_step> times per step,
e.g. k_halo_unpack: once per slab) ended <steps> steps before the end, and ends with the last kernel of the trace.  If one
host thread driving 8 slabs could not keep the device fed, it would show here as idle gaps between kernels."""
import csv
import json
import sys

rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
steps, marker, per_step = int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
rows.sort()
marks = [e for s, e, n in rows if marker in n]
lo = marks[-(steps * per_step) - 1]
t1 = max(r[1] for r in rows)
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
print(json.dumps({"steps_in_window": steps, "window_ms": round(span / 1e6, 3), "ms_per_step": round(span / 1e6 / steps, 3), "busy_ms": round(busy / 1e6, 3),
                  "idle_ms": round((span - busy) / 1e6, 3), "idle_fraction": round((span - busy) / span, 5), "gaps": len(gaps),
                  "largest_gaps_us": [round(g / 1e3, 1) for g in gaps[:5]], "kernels_in_window": len(iv)}))
