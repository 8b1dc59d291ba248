#!/usr/bin/env python3
"""Condense a gpurun_out/<tag>/ profile directory (scripts/gpu_profile.sh) into profiles/<name>/:
kernel_stats.csv (rocprofv3 --kernel-trace --stats), pmc_summary.csv (per-kernel counter means) and
bench.json.  usage: summarize_profile.py gpurun_out/<tag> profiles/<name>"""
import collections, csv, glob, json, os, shutil, sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
latest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
shutil.copy(latest(f"{src}/trace/runc/*_kernel_stats.csv"), f"{dst}/kernel_stats.csv")
shutil.copy(f"{src}/bench.json", f"{dst}/bench.json")
rows = []
for d in sorted(glob.glob(f"{src}/pmc*/")):
    f = latest(f"{d}runc/*_counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        for c, x in v.items():
            rows.append((k, c, len(x), sum(x) / len(x)))
with open(f"{dst}/pmc_summary.csv", "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Kernel_Name", "Counter_Name", "Dispatches", "Mean_Value_Per_Dispatch"])
    for r in sorted(rows):
        w.writerow(r)
print(open(f"{dst}/kernel_stats.csv").read()[:1500])
