#!/bin/bash
# one PMC pass over a run: scripts/pmc_one.sh <tag> "<counters>" <script.py> [args]; prints per-kernel means
TAG=$1; shift; CTR=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --pmc $CTR --output-format csv -d $OUT/pmc -- python3 "$@" > $OUT/pmc.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/pmc/**/*counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"].split("(")[0][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print(k.ljust(48), " ".join("%s=%.3g"%(c,sum(x)/len(x)) for c,x in sorted(v.items())))
PY
