#!/bin/bash
# walk_ms of small launches: the level-synchronous split walk (automatic factor) against the one-wave
# assembly loop (BH_WALK_SPLIT=1), Plummer
for n in 16384 65536 131072 262144; do
  for sp in 0 1 2; do
    BH_WALK_SPLIT=$sp python bench.py --no-cpu-baseline --no-secondary --n-bodies $n --steps 50 --warmup 5 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('n=$n split=$sp ms/step %.4f walk %.4f build %.4f' % (j['ms_per_step'], j['walk_ms'], j['build_ms']))"
  done
done
