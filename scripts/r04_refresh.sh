#!/bin/bash
# Everything that is stamped with the digest of the device sources, again, after a change to csrc/ or include/bhgpu.h
# (run in the development container, from the repository root; two gpurun calls):
#   experiments build -> smoke + the five profile blocks of r04_profiles.sh -> summaries into profiles/ -> bench line
# GPURUN: the gpurun client (default /usr/local/graft/bin/gpurun).
set -e
G=${GPURUN:-/usr/local/graft/bin/gpurun}
bash scripts/build_variants.sh exp:"-DBHGPU_EXPERIMENTS" | tail -1
$G --timeout 1200 -- 'python -c "import __graft_entry__ as g; g.smoke()" && bash scripts/r04_profiles.sh c3 > /dev/null 2>&1 && bash scripts/r04_profiles.sh clock > /dev/null 2>&1 && bash scripts/r04_profiles.sh f64 > /dev/null 2>&1 && bash scripts/r04_profiles.sh exact > /dev/null 2>&1 && bash scripts/r04_profiles.sh others > /dev/null 2>&1; echo done' | tail -3
python scripts/summarize_profile.py gpurun_out/r04_c3 profiles/r04_final > /dev/null
for t in c4:C4 c5:C5 c2:C2 f64:F64 exact:EXACT; do
  python scripts/summarize_profile.py gpurun_out/r04_${t%%:*} profiles/r04_${t%%:*} --tag ${t##*:} > /dev/null
done
for f in walk_clock_c3.json walk_clock_c5.json timeline_c3.txt timeline_c5.txt; do cp gpurun_out/r04_clock/$f profiles/r04_final/$f; done
cp gpurun_out/r04_clock/walk_clock_c3.json profiles/latest_walk_clock.json
grep -h source_digest profiles/latest_walk_traffic*.json | sort | uniq -c
$G --timeout 900 -- 'timeout -k 10 600 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -c 100 gpurun_out/bench_final.err' | tail -1
cp gpurun_out/bench_final.json profiles/r04_final/bench.json
