#!/bin/bash
# kernel-level trace of the bit-exact fp64 mode (scripts/exact_timing.py): which kernel is its build made of?
mkdir -p gpurun_out/r02_ex
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_ex/trace -- python3 scripts/exact_timing.py > gpurun_out/r02_ex/trace.log 2>&1
python3 - <<PY
import csv, glob
f = max(glob.glob("gpurun_out/r02_ex/trace/*/*_kernel_stats.csv"))
for r in list(csv.DictReader(open(f)))[:12]:
    print(r["Name"].split("(")[0][:60], r["Calls"], "avg %.1f us" % (float(r["AverageNs"]) / 1e3), "min %.1f max %.1f" % (float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
