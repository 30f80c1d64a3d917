"""gpu_busy_from_trace.py <kernel_trace.csv> <steps> [marker_kernel markers_per_step] - how much of the wall time of the
last <steps> time steps of a run was the GPU idle?

Union of all kernel intervals (device copies show up as __amd_rocclr_copyBuffer kernels) of a rocprofv3 --kernel-trace.
The window is the last <steps> steps and ends with the last kernel of the trace.  Without a marker the steps are cut by
tools/trace_steps.py (the boundary-plane launch that opens a step of the slab path) and the window starts where the
<steps>-th step from the end starts; with one it starts when the marker kernel (one that runs <markers_per_step> times
per step) ended <steps> steps before the end.  If one host thread driving 8 slabs could not keep the device fed, it
would show here as idle gaps between kernels."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from trace_steps import load_rows, step_starts  # noqa: E402

rows_d = load_rows(sys.argv[1])
rows = [(r["s"], r["e"], r["n"]) for r in rows_d]
steps = int(sys.argv[2])
if len(sys.argv) > 4:
    marker, per_step = sys.argv[3], int(sys.argv[4])
    marks = [e for s, e, n in rows if marker in n]
    lo = marks[-(steps * per_step) - 1]
else:
    st = step_starts(rows_d)
    if len(st) < steps:
        print(f"gpu_busy_from_trace: {len(st)} steps in the trace, {steps} asked for", file=sys.stderr)
        sys.exit(3)
    lo = rows_d[st[-steps]]["s"]
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
