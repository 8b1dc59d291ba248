#!/bin/bash
# per-kernel durations of a step: scripts/trace_step.sh <tag> [bench args]   (on the GPU box)
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -- python3 $R/bench.py --no-secondary --no-cpu-baseline "$@" > $R/gpurun_out/$TAG.log 2>&1
python3 - <<PY
import csv, glob
f=glob.glob("$R/gpurun_out/$TAG/*/*_kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:18]:
    print("%-64s calls %5s avg %7.1f us" % (r["Name"][:64], r["Calls"], float(r["AverageNs"])/1e3))
PY
