"""summarize_sq.py <tag> - gpurun_out/prof_sq_<tag>/{sq,tcc}/pmc_counter_collection.csv -> profiles/<tag>_cfg3_pmc_sq_lds_tcc.json:
per kernel (its largest launches only) the mean of every counter, plus derived L2 hit rate, VALU
instructions per wave and the share of wave cycles spent waiting."""
import collections, csv, json, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_sq_{tag}")
out = collections.defaultdict(dict)
for sub in ("sq", "tcc"):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(os.path.join(src, sub, "pmc_counter_collection.csv"))):
        d[r["Kernel_Name"]][r["Counter_Name"]].append((int(r.get("Grid_Size", 0) or 0), float(r["Counter_Value"])))
    for k, cs in d.items():
        if not any(t in k for t in ("k_collide", "k_tridiag", "k_phi_efield_x2", "fft_rtc")):
            continue
        for cn, v in cs.items():
            g = max(x[0] for x in v)
            big = [x[1] for x in v if x[0] == g]
            out[k.split("(")[0].replace("void ", "")][cn] = sum(big) / len(big)
for k, c in out.items():
    if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum"):
        c["L2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
    if c.get("SQ_WAVES"):
        c["VALU_insts_per_wave"] = round(c.get("SQ_INSTS_VALU", 0) / c["SQ_WAVES"], 1)
        c["LDS_insts_per_wave"] = round(c.get("SQ_INSTS_LDS", 0) / c["SQ_WAVES"], 2)
path = os.path.join(root, "profiles", f"{tag}_cfg3_pmc_sq_lds_tcc.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps({k: {a: b for a, b in v.items() if a in ("SQ_WAVES", "L2_hit_rate", "VALU_insts_per_wave", "LDS_insts_per_wave", "SQ_LDS_BANK_CONFLICT")} for k, v in out.items()}, indent=1))
