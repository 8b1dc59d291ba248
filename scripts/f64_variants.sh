#!/bin/bash
# BH_PRECISION_F64 A/B on the GPU box: scripts/f64_ab.py for the product library and every named build variant
# (scripts/build_variants.sh name:"flags" ... first).  usage: scripts/f64_variants.sh [variant ...]
cd "$(dirname "$0")/.."
for v in "" "$@"; do
  if [ -n "$v" ]; then export BHGPU_LIB_OPT_IN=1 BHGPU_LIB=$PWD/gpu-nbody-simulation_amd/build/libbhgpu_$v.so; fi
  timeout -k 10 200 python3 scripts/f64_ab.py 2>&1 | tail -1
done
