#!/bin/bash
# build_ms: bucket sort (one counting pass by splitters + in-LDS bucket sorts, default) against the five LSD
# passes (BH_SORT_BUCKET=0); INIT/DRIFT choose the workload (drift_cells = 1: every body changes cell)
for n in ${SIZES:-16384 65536 131072 262144 524288 1048576}; do
  for g in 1 0; do
    BH_SORT_BUCKET=$g python bench.py --no-cpu-baseline --n-bodies $n --steps 50 --warmup 5 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); g=j['build_groups_ms']; d=j['dynamic']
print('n=$n bucket=$g ms/step %.4f walk %.4f build %.4f | keys %.4f sort %.4f scan %.4f nodes %.4f | dynamic leg ms/step %.4f build %.4f' % (j['ms_per_step'], j['walk_ms'], j['build_ms'], g['keys'], g['sort'], g['scan'], g['nodes'], d['ms_per_step'], d['build_ms']))"
  done
done
