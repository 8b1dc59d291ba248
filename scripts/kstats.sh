#!/bin/bash
# per-kernel average durations (rocprofv3 --kernel-trace --stats) of a run: scripts/kstats.sh <tag> <script.py> [args]
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 "$@" > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/trace/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(r["Name"].split("(")[0][:60].ljust(60), r["Calls"].rjust(5), "%.2f"%(float(r["AverageNs"])/1e3))
PY
