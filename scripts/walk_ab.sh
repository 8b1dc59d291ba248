#!/bin/bash
# A/B of walk-kernel build variants on the GPU box: for every libbhgpu_<variant>.so under
# gpu-nbody-simulation_amd/build/ (python -m gpu_nbody_simulation_amd.build --variant NAME -D...) and for
# the product library: the asm-vs-portable bitwise test, then walk_ms / ms_per_step of the bench workload.
# usage: scripts/walk_ab.sh <outdir> [bench args...]
OUT=$1; shift
mkdir -p $OUT
PKG=gpu-nbody-simulation_amd
for lib in $PKG/libbhgpu.so $PKG/build/libbhgpu_*.so; do
  [ -f "$lib" ] || continue
  tag=$(basename $lib .so)
  BHGPU_LIB_OPT_IN=1 BHGPU_LIB=$PWD/$lib timeout -k 10 300 python -m pytest tests/test_gpu_fp32.py -x -q -k "asm_walk or split_walk or bucket_mode" > $OUT/$tag.pytest.log 2>&1
  echo "$tag pytest rc=$? $(tail -1 $OUT/$tag.pytest.log)"
  for rep in 1 2; do
    BHGPU_LIB_OPT_IN=1 BHGPU_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary "$@" > $OUT/$tag.$rep.json 2> $OUT/$tag.err || { echo "$tag bench failed"; tail -3 $OUT/$tag.err; break; }
    python - <<PY
import json
j = json.load(open("$OUT/$tag.$rep.json"))
print("$tag", "ms/step %.4f  walk %.4f  build %.4f" % (j["ms_per_step"], j["walk_ms"], j["build_ms"]))
PY
  done
done
