"""step_timeline.py <kernel_trace.csv> [marker] - the kernels of the LAST full step of a traced run, in start order:
offset from the step's first kernel, duration, stream, and the gap since the previous kernel ended on the same stream.
Steps are cut by tools/trace_steps.py (at the boundary-plane launch that opens a step of the slab path); with a
`marker` argument a step runs from one kernel whose name contains it to the next instead (single-context runs:
k_collide_wall).  Shows where a step's time goes between its big kernels."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from trace_steps import load_rows, short_name, step_starts  # noqa: E402


def timeline(rows, marker=None):
    starts = [i for i, r in enumerate(rows) if marker in r["n"]] if marker else step_starts(rows)
    if len(starts) < 2:
        return None
    a, b = starts[-2], starts[-1]
    step = rows[a:b]
    t0 = step[0]["s"]
    last_end = {}
    out = []
    for r in step:
        gap = (r["s"] - last_end[r["q"]]) / 1e3 if r["q"] in last_end else None
        last_end[r["q"]] = max(last_end.get(r["q"], 0), r["e"])
        out.append({"at_ms": round((r["s"] - t0) / 1e6, 4), "ms": round((r["e"] - r["s"]) / 1e6, 4), "stream": r["q"], "gap_us": None if gap is None else round(gap, 1), "kernel": short_name(r["n"])[:60]})
    return {"step_ms": round((rows[b]["s"] - t0) / 1e6, 4), "kernels": out}


if __name__ == "__main__":
    res = timeline(load_rows(sys.argv[1]), sys.argv[2] if len(sys.argv) > 2 else None)
    if res is None:
        print("step_timeline: fewer than two step starts in the trace", file=sys.stderr)
        sys.exit(3)
    print(json.dumps(res, indent=0))
