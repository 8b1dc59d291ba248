#!/usr/bin/env python3
"""Condense a gpurun_out/<tag>/ profile directory (scripts/gpu_profile.sh) into profiles/<name>/:
kernel_stats.csv (rocprofv3 --kernel-trace --stats), pmc_summary.csv (per-kernel counter means) and
bench.json.  usage: summarize_profile.py gpurun_out/<tag> profiles/<name> [--tag C2|C4|C5|F64|EXACT] [--no-latest]
--tag: the profile is of another configuration than the headline: its traffic goes to profiles/latest_walk_traffic_<tag>.json,
which bench.py reads for that leg (other_configs.<tag>.roofline.traffic / secondary_f64.roofline.traffic)."""
import collections, csv, glob, json, os, shutil, subprocess, sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gpu_nbody_simulation_amd.build import source_digest  # noqa: E402

src, dst = sys.argv[1], sys.argv[2]
update_latest = "--no-latest" not in sys.argv[3:]          # other configurations than the headline: keep profiles/latest_*
tag = sys.argv[sys.argv.index("--tag") + 1] if "--tag" in sys.argv else None
os.makedirs(dst, exist_ok=True)
latest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
shutil.copy(latest(f"{src}/trace/runc/*_kernel_stats.csv"), f"{dst}/kernel_stats.csv")
if os.path.exists(f"{src}/bench.json"):
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
# HBM traffic of the walk kernel per launch (MI355X_MICROARCH.md, HBM/rocprofv3): FETCH_SIZE and
# WRITE_SIZE are in KB, collected in separate --pmc passes; on gfx950 FETCH_SIZE reports exactly 1/2
# of the bytes read -- calibrated for THIS kernel's access pattern (wave-uniform 64-byte scalar
# loads) with scripts/calib/fetch_calib.hip: ratio 0.50003 -- so reads = 2 * FETCH_SIZE.
# the product walk of the run: the hand-scheduled kernel, or the C++ loop where the engine had to fall back to it
prefix = {"F64": "void bh::walk_f64_kernel", "EXACT": "void bh::walk_exact_kernel"}.get(tag, "void bh::walk_fast_kernel")
names = collections.Counter()
for (k, c, n, v) in rows:
    if k.startswith(prefix) and c == "FETCH_SIZE":
        names[k] += n
# the product walk of the run = the instantiation with the most dispatches (one wave per group at C3-C5, four at C2)
wname = names.most_common(1)[0][0] if names else None
walk = {c: v for (k, c, n, v) in rows if k == wname}
calib = {}
for f in glob.glob(f"{src}/calib/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("k_") and r["Counter_Name"] == "FETCH_SIZE":
            calib[r["Kernel_Name"].split("(")[0]] = float(r["Counter_Value"]) * 1024 / 2**30
if "FETCH_SIZE" in walk and "WRITE_SIZE" in walk:
    t = {"kernel": wname.split("(")[0].replace("void bh::", ""), "fetch_size_kb": walk["FETCH_SIZE"], "write_size_kb": walk["WRITE_SIZE"],
         "fetch_correction": 2.0, "traffic_bytes": 2.0 * walk["FETCH_SIZE"] * 1024 + walk["WRITE_SIZE"] * 1024,
         "l2_hit_rate": walk.get("TCC_HIT_sum", 0) / max(1.0, walk.get("TCC_HIT_sum", 0) + walk.get("TCC_MISS_sum", 0)),
         "calibration_fetch_size_over_true_bytes": calib, "source": os.path.basename(dst),
         # what the profile was measured on: bench.py drops `traffic` when the kernels have changed since
         "source_digest": source_digest(),
         "git_head": subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip() or None}
    json.dump(t, open(f"{dst}/walk_traffic.json", "w"), indent=1)
    if tag:
        json.dump(t, open(os.path.join(os.path.dirname(dst.rstrip("/")), f"latest_walk_traffic_{tag}.json"), "w"), indent=1)
    elif update_latest:
        json.dump(t, open(os.path.join(os.path.dirname(dst.rstrip("/")), "latest_walk_traffic.json"), "w"), indent=1)
    print(json.dumps(t))
print(open(f"{dst}/kernel_stats.csv").read()[:1800])
