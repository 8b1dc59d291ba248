#!/bin/bash
# per-kernel averages of run_steps.py for the product library and every named variant: scripts/kstats_lib.sh <grep-pattern> [variant ...]
PAT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in "" "$@"; do
  OUT=gpurun_out/kl_${v:-product}
  rm -rf $OUT/trace; mkdir -p $OUT
  if [ -n "$v" ]; then export BHGPU_LIB_OPT_IN=1 BHGPU_LIB=$PWD/gpu-nbody-simulation_amd/build/libbhgpu_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 scripts/run_steps.py ${RUN_ARGS:---steps 30} > $OUT/trace.log 2>&1
  echo "== ${v:-product}: $(grep '^{' $OUT/trace.log)"
  python3 - <<PY
import csv,glob,re
f=glob.glob("$OUT/trace/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if re.search("$PAT", r["Name"]): print("   ", r["Name"].split("(")[0][:60].ljust(60), r["Calls"].rjust(5), "%.2f"%(float(r["AverageNs"])/1e3))
PY
done
