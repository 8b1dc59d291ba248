#!/bin/bash
# kernel-trace only (fast): scripts/gpu_trace.sh <tag> [bench args]
set -e
TAG=$1; shift
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --no-secondary "$@" > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log
python3 - <<PY
import csv, glob
f=glob.glob("$OUT/trace/*/*_kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print("%-60s calls %5s avg %9.1f us  %5.1f%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
