#!/bin/bash
# A/B of the packed (key|index) radix passes (BH_SORT_PACK=0: separate key and index arrays)
for v in 1 0 1 0; do
  BH_SORT_PACK=$v python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('pack=$v ms/step %.4f walk %.4f build %.4f' % (j['ms_per_step'], j['walk_ms'], j['build_ms']))"
done
