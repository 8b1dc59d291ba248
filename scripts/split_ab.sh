#!/bin/bash
# walk_ms against the split factor of the level-synchronous walk (BH_WALK_SPLIT: 0 = automatic, 1 = the
# one-wave depth-first assembly loop), Plummer
for n in ${SIZES:-65536 131072 262144 524288 1048576}; do
  for sp in ${SPLITS:-0 1 2 4 8}; do
    BH_WALK_SPLIT=$sp python bench.py --no-cpu-baseline --no-secondary --n-bodies $n --steps 50 --warmup 5 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('n=$n split=$sp ms/step %.4f walk %.4f build %.4f' % (j['ms_per_step'], j['walk_ms'], j['build_ms']))"
  done
done
