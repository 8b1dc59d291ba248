#!/bin/bash
# (the variants compared here are compiled only into an experiments build: `python -m gpu_nbody_simulation_amd.build --variant exp -DBHGPU_EXPERIMENTS`, then run with BHGPU_LIB_OPT_IN=1 BHGPU_LIB=gpu-nbody-simulation_amd/build/libbhgpu_exp.so)
# A/B of walk variants in one gpurun call: prints ms_per_step / build / walk for each setting
for cfg in "0 0" "2 0" "1 0" "0 0" "2 0"; do
  set -- $cfg
  BH_WALK_PIPE=$1 BH_WALK_XCD=$2 python3 bench.py --no-cpu-baseline --steps 30 --warmup 5 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('pipe=$1 xcd=$2  ms/step %.3f  build %.3f  walk %.3f  nodes %d inter %.2f' % (d['ms_per_step'], d['build_ms'], d['walk_ms'], d['n_nodes'], d['interactions_per_body']))"
done
