#!/usr/bin/env python3
"""Condense a gpurun_out/prof_round directory (see tools/prof_round.sh) into profiles/<tag>_*.{csv,json}."""
import collections, csv, glob, json, os, re, sys

base = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_round"
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
os.makedirs("profiles", exist_ok=True)


def short(name):
    head = name.split("(")[0] if not name.startswith("(") and not name.startswith("void (") else re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
    return head.replace("void ", "").split("::")[-1].strip()


newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)   # gpurun merges into gpurun_out/: older runs may still be there
src = newest(f"{base}/trace/*/*_kernel_stats.csv")
with open(f"profiles/{tag}_bench_kernel_stats.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["Kernel", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in csv.DictReader(open(src)):
        w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
out = {}
for d in ("fetch", "write", "sq"):
    fs = glob.glob(f"{base}/{d}/*/*_counter_collection.csv")
    if not fs:
        continue
    fs = [max(fs, key=os.path.getmtime)]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if "k_edt" in r["Kernel_Name"]:
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in agg:
        for c, v in agg[k].items():
            out.setdefault(k, {})[c] = sum(v) / len(v)
out["_note"] = ("per-launch averages; FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them (separate --pmc passes). "
                "On gfx950 FETCH_SIZE is half of the bytes read, for 4-byte and 16-byte loads per lane alike, and WRITE_SIZE is "
                "exact (tools/fetch_calib.hip: 0.500 / 1.000 of a streamed GiB); HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE.")
json.dump(out, open(f"profiles/{tag}_edt_pmc.json", "w"), indent=1, sort_keys=True)
print(json.dumps({k: {c: v for c, v in d.items() if c in ("FETCH_SIZE", "WRITE_SIZE")} for k, d in out.items() if isinstance(d, dict)}, indent=1))
