#!/bin/bash
for n in 16384 65536 262144 1048576 4194304 16777216; do
  for init in uniform plummer; do
    python3 bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3 --n-bodies $n --init $init 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('N=%9d %-8s ms/step %8.3f  build %7.3f  walk %8.3f  %8.1f M body-steps/s  inter/body %.1f nodes %d' % ($n, '$init', d['ms_per_step'], d['build_ms'], d['walk_ms'], d['value']/1e6, d['interactions_per_body'], d['n_nodes']))"
  done
done
