#!/bin/bash
# The other BASELINE configurations on one MI355X: ms/step, build, walk
run() { python bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$*', '| ms/step %.4f  %.3f G body-steps/s  build %.4f walk %.4f  int/body %.1f' % (j['ms_per_step'], j['value']/1e9, j['build_ms'], j['walk_ms'], j['interactions_per_body']))"; }
run --n-bodies 65536 --init uniform --steps 1000 --warmup 20
run --n-bodies 65536 --init plummer --steps 200 --warmup 20
run --n-bodies 262144 --init plummer --steps 50 --warmup 5
run --n-bodies 4194304 --init plummer --steps 10 --warmup 2
run --n-bodies 4194304 --init uniform --steps 10 --warmup 2
run --n-bodies 16777216 --init plummer --theta 0.3 --precision mixed --steps 5 --warmup 1
run --n-bodies 16777216 --init plummer --steps 5 --warmup 1
run --n-bodies 16777216 --init uniform --steps 5 --warmup 1
