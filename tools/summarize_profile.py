"""Condenses gpurun_out/prof_<tag>/ (tools/profile.sh) into profiles/<tag>_*: the kernel-stats
CSV, a PMC summary with the gfx950 FETCH_SIZE correction calibrated on tools/pmc_calib, and
profiles/pmc_traffic.json (HBM bytes per launch of the dominant kernel, read by bench.py).

    python tools/summarize_profile.py TAG [WORKLOAD]        # gpurun_out/prof_TAG -> profiles/

summarize(tag, wl, src, dst) is the same with the two directories given (tests/test_tools_cpu.py runs it on a synthetic
profile directory: the calibration factor, the choice of the steady-state sweep among the instantiations and the per-step
table are arithmetic this repository's roofline.traffic rests on)."""
import collections, csv, json, os, shutil, sys


def agg(path):
    """mean counter value per kernel over its LARGEST launches only (the bench also runs the same
    kernels on a 16x16x257 replica for its start state; those launches must not dilute the mean)"""
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"]].append((int(r.get("Grid_Size", 0) or 0), float(r["Counter_Value"])))
    out = {}
    for k, v in d.items():
        g = max(x[0] for x in v)
        big = [x[1] for x in v if x[0] == g]
        out[k] = sum(big) / len(big)
    return out


def steady_sweep(kernels):
    """the steady-state sweep is the PULL = true instantiation of k_collide_bulk (first template flag; the second, since round 4,
    says whether E is formed from phi)"""
    return next(k for k in kernels if "k_collide_bulk<" in k and k.split("k_collide_bulk<", 1)[1].split(",")[1].strip() == "true")


def summarize(tag, wl, src, dst):
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, f"{tag}_{wl}_kernel_stats.csv"))
    cf, cw = agg(os.path.join(src, "calib_fetch", "pmc_counter_collection.csv")), agg(os.path.join(src, "calib_write", "pmc_counter_collection.csv"))
    copy_name = next(k for k in cf if k.startswith("k_copy8("))
    shift_name = next(k for k in cf if k.startswith("k_copy8_shift("))
    true_kib = (1 << 30) * 8 / 1024
    fetch_factor = true_kib / cf[copy_name]          # 2.0 on gfx950: FETCH_SIZE tallies 128-B requests at 64 B
    write_factor = true_kib / cw[copy_name]
    pf, pw = agg(os.path.join(src, "pmc_fetch", "pmc_counter_collection.csv")), agg(os.path.join(src, "pmc_write", "pmc_counter_collection.csv"))
    out = {"calibration": {"kernel": "tools/pmc_calib k_copy8 (8 B/lane, 8 GiB read + 8 GiB write)",
                           "FETCH_SIZE_KiB": cf[copy_name], "WRITE_SIZE_KiB": cw[copy_name], "true_KiB": true_kib,
                           "fetch_factor": fetch_factor, "write_factor": write_factor,
                           "misaligned_by_one_element_fetch_ratio": cf[shift_name] / cf[copy_name]},
           "kernels": {}}
    for k in sorted(set(pf) | set(pw)):
        f, w = pf.get(k, 0.0), pw.get(k, 0.0)
        out["kernels"][k] = {"FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB_raw": w,
                             "hbm_read_bytes": f * 1024 * fetch_factor, "hbm_write_bytes": w * 1024 * write_factor,
                             "hbm_bytes_per_launch": f * 1024 * fetch_factor + w * 1024 * write_factor}
    json.dump(out, open(os.path.join(dst, f"{tag}_{wl}_pmc_summary.json"), "w"), indent=1)
    bulk = steady_sweep(out["kernels"])
    tpath = os.path.join(dst, "pmc_traffic.json")
    t = json.load(open(tpath)) if os.path.exists(tpath) else {}
    t[wl] = {"kernel": bulk, "hbm_bytes_per_launch": out["kernels"][bulk]["hbm_bytes_per_launch"], "round": tag,
             "note": "(FETCH_SIZE*fetch_factor + WRITE_SIZE)*1024, separate --pmc passes, calibrated on tools/pmc_calib"}
    # what one STEADY-STATE STEP moves, kernel by kernel: the kernels between two consecutive launches of the sweep in the
    # per-dispatch trace of pass 1 (the last full step), each with the counter bytes of its launch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from trace_steps import load_rows  # noqa: E402

    trace = os.path.join(src, "trace", "trace_kernel_trace.csv")
    if os.path.exists(trace):
        rows = load_rows(trace)
        gmax = max(r["gx"] for r in rows if r["n"] == bulk)
        marks = [i for i, r in enumerate(rows) if r["n"] == bulk and r["gx"] == gmax]
        step = collections.Counter(r["n"] for r in rows[marks[-2]:marks[-1]])
        table = [{"kernel": k, "launches_per_step": n, "hbm_bytes_per_launch": out["kernels"].get(k, {}).get("hbm_bytes_per_launch")} for k, n in sorted(step.items())]
        t[wl]["step"] = {"kernels": table, "hbm_bytes_per_step": sum(e["launches_per_step"] * (e["hbm_bytes_per_launch"] or 0.0) for e in table),
                         "kernels_without_counters": [e["kernel"] for e in table if e["hbm_bytes_per_launch"] is None],
                         "note": "kernels of the last full step of the kernel trace x the counter bytes of their launches (same profile); bench.py prints the sum as config.step_traffic_bytes"}
    json.dump(t, open(tpath, "w"), indent=1)
    calib_log = os.path.join(src, "calib_fetch.log")
    if os.path.exists(calib_log):
        shutil.copy(calib_log, os.path.join(dst, f"{tag}_pmc_calib.log"))
    return out, bulk, t[wl]


if __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    wl = sys.argv[2] if len(sys.argv) > 2 else "cfg3"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out, bulk, _ = summarize(tag, wl, os.path.join(root, "gpurun_out", f"prof_{tag}"), os.path.join(root, "profiles"))
    print(json.dumps(out["calibration"], indent=1)); print(bulk, out["kernels"][bulk])
