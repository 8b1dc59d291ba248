#!/bin/bash
# walk_ms of the level-synchronous walk of small launches: the assembly chunk loop (default) against the
# C++ chunk loop (BH_WALK_ASM=0), Plummer and uniform
for init in plummer uniform; do
for n in 16384 65536 100000; do
  for asm in 1 0; do
    BH_WALK_ASM=$asm python bench.py --no-cpu-baseline --no-secondary --init $init --n-bodies $n --steps 200 --warmup 10 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$init n=$n asm=$asm ms/step %.4f walk %.4f build %.4f' % (j['ms_per_step'], j['walk_ms'], j['build_ms']))"
  done
done
done
