#!/bin/bash
# Does the walk care where its quads are cached?  Experiments build (scripts/build_variants.sh exp:"-DBHGPU_EXPERIMENTS",
# WALK_ONLY=0), default round-robin block -> XCD placement against the XCD-contiguous one (BH_WALK_XCD=1): kernel time
# (rocprofv3 --kernel-trace --stats) and L2 hit / miss + wave wait counters of the walk kernel (separate --pmc pass).
OUT=gpurun_out/${1:-xcd_probe}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export BHGPU_LIB_OPT_IN=1 BHGPU_LIB=$PWD/gpu-nbody-simulation_amd/build/libbhgpu_exp.so
for x in 0 1; do
  export BH_WALK_XCD=$x
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace$x -- python3 scripts/run_steps.py --steps 20 --warmup 3 > $OUT/trace$x.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc$x -- python3 scripts/run_steps.py --steps 4 --warmup 2 > $OUT/pmc$x.log 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/trace$x/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "walk_fast" in r["Name"]:
        print("BH_WALK_XCD=$x", r["Name"][:60], "calls", r["Calls"], "avg us %.2f" % (float(r["AverageNs"]) / 1e3))
f = glob.glob("$OUT/pmc$x/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "walk_fast" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("BH_WALK_XCD=$x", " ".join("%s=%.4g" % (k, sum(v) / len(v)) for k, v in sorted(acc.items())))
PY
done
