"""summarize_sq.py <tag> [label] - gpurun_out/prof_sq_<tag>_<label>/{sq,lds,tcc[,fetch,write]}/pmc_counter_collection.csv ->
gpurun_out/<tag>_<label>_pmc_sq_lds_tcc.json (copy it to profiles/): per kernel (its largest launches only) the mean of every counter, plus derived
L2 hit rate, instructions per wave, the share of wave cycles spent waiting / issuing, and - when the FETCH_SIZE / WRITE_SIZE
passes were made (EKPNP_PROFILE_HBM=1) - HBM bytes per launch with the gfx950 FETCH_SIZE factor 2.0 calibrated on
tools/pmc_calib (profiles/r03_pmc_calib.log)."""
import collections, csv, json, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
label = sys.argv[2] if len(sys.argv) > 2 else "cfg3"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_sq_{tag}_{label}")
KEEP = ("k_collide", "k_tridiag", "k_slab_", "k_phi_efield", "fft_rtc", "k_fft_", "k_halo", "k_ghost")
out = collections.defaultdict(dict)
for sub in ("sq", "lds", "tcc", "fetch", "write"):
    path = os.path.join(src, sub, "pmc_counter_collection.csv")
    if not os.path.exists(path):
        continue
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"]][r["Counter_Name"]].append((int(r.get("Grid_Size", 0) or 0), float(r["Counter_Value"])))
    for k, cs in d.items():
        if not any(t in k for t in KEEP):
            continue
        for cn, v in cs.items():
            g = max(x[0] for x in v)
            big = [x[1] for x in v if x[0] == g]
            e = out[k.split("(")[0].replace("void ", "")]
            e[cn] = sum(big) / len(big)
            e["grid_size"] = g
for k, c in out.items():
    if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum"):
        c["L2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
    w = c.get("SQ_WAVES")
    if w:
        c["VALU_insts_per_wave"] = round(c.get("SQ_INSTS_VALU", 0) / w, 1)
        for a, b in (("SQ_INSTS_LDS", "LDS_insts_per_wave"), ("SQ_INSTS_VMEM_RD", "VMEM_RD_insts_per_wave"), ("SQ_INSTS_VMEM_WR", "VMEM_WR_insts_per_wave")):
            if a in c:
                c[b] = round(c[a] / w, 2)
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:  # all four count quad-cycles; WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES (MI355X_MICROARCH.md, PMC slots)
        for a, b in (("SQ_WAIT_ANY", "frac_wave_cycles_parked_waitcnt_or_barrier"), ("SQ_WAIT_INST_ANY", "frac_wave_cycles_issue_stalled"),
                     ("SQ_ACTIVE_INST_ANY", "frac_wave_cycles_issuing"), ("SQ_ACTIVE_INST_VALU", "frac_wave_cycles_issuing_valu"),
                     ("SQ_ACTIVE_INST_LDS", "frac_wave_cycles_issuing_lds"), ("SQ_WAIT_INST_LDS", "frac_wave_cycles_lds_issue_stall"),
                     ("SQ_ACTIVE_INST_VMEM", "frac_wave_cycles_issuing_vmem")):
            if a in c:
                c[b] = round(c[a] / wc, 4)
    if c.get("SQ_LDS_IDX_ACTIVE"):
        c["lds_bank_conflict_share_of_lds_cycles"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
    if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
        c["hbm_read_bytes"] = c.get("FETCH_SIZE", 0.0) * 1024 * 2.0
        c["hbm_write_bytes"] = c.get("WRITE_SIZE", 0.0) * 1024
        c["hbm_bytes_per_launch"] = c["hbm_read_bytes"] + c["hbm_write_bytes"]
# written under gpurun_out/ (what comes back from the GPU box; the raw per-dispatch CSVs are too large to); copy to profiles/ to keep
path = os.path.join(root, "gpurun_out", f"{tag}_{label}_pmc_sq_lds_tcc.json")
json.dump(out, open(path, "w"), indent=1)
show = ("SQ_WAVES", "L2_hit_rate", "VALU_insts_per_wave", "LDS_insts_per_wave", "frac_wave_cycles_parked_waitcnt_or_barrier",
        "frac_wave_cycles_issue_stalled", "frac_wave_cycles_issuing", "frac_wave_cycles_issuing_valu", "frac_wave_cycles_lds_issue_stall",
        "lds_bank_conflict_share_of_lds_cycles", "hbm_bytes_per_launch", "grid_size")
print(json.dumps({k: {a: b for a, b in v.items() if a in show} for k, v in out.items()}, indent=1))
