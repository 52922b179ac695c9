"""rocprofv3 --pmc passes (one counter set per run, MI355X_MICROARCH.md 'HBM') -> the per-launch traffic JSON bench.py reads.
usage: pmc_summary.py <dir with pass_*/..._counter_collection.csv> <kernel substring> <algorithmic bytes> <launch text> <out.json>"""
import csv, glob, json, os, sys
root, kname, alg, launch, out = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4], sys.argv[5]
tot, cnt = {}, {}
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if kname not in r["Kernel_Name"]:
            continue
        c = r["Counter_Name"]
        tot[c] = tot.get(c, 0.0) + float(r["Counter_Value"])
        cnt[c] = cnt.get(c, set()); cnt[c].add(r["Dispatch_Id"])
avg = {c: tot[c] / len(cnt[c]) for c in tot}
res = {"_comment": "rocprofv3 --pmc, separate passes (FETCH_SIZE | WRITE_SIZE | SQ/GRBM set), averages per launch of the named kernel. "
                   "FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); units KiB.",
       "kernel": kname, "launch": launch, "launches_averaged": {c: len(cnt[c]) for c in cnt}}
for c, v in avg.items():
    res[c + ("_KiB_raw" if c in ("FETCH_SIZE", "WRITE_SIZE") else "")] = round(v, 2)
if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
    res["hbm_bytes_per_launch"] = int(avg["FETCH_SIZE"] * 1024 * 2 + avg["WRITE_SIZE"] * 1024)
res["algorithmic_bytes_per_launch"] = int(alg)
if "GRBM_GUI_ACTIVE" in avg and "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
    res["mfma_busy_frac_of_simd_cycles"] = round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (avg["GRBM_GUI_ACTIVE"] / 8 * 1024), 3)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
