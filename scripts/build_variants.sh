#!/bin/bash
# Build A/B variants of libbhgpu (gpu-nbody-simulation_amd/build/libbhgpu_<name>.so) in parallel.
# usage: scripts/build_variants.sh name1:"-DFOO=1 -DBAR=2" name2:"-DBAZ=1" ...
# WALK_ONLY=1 recompiles only bh_walk_fast.hip (the engine object of the product build is reused: faster, but
# bh_build_info() then still says "product").
cd "$(dirname "$0")/.."
PKG=gpu-nbody-simulation_amd
python -m gpu_nbody_simulation_amd.build > /dev/null || exit 1
DIGEST=$(python -c "from gpu_nbody_simulation_amd.build import source_digest; print(source_digest())")
pids=()
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  (
    d=$PKG/build/$name; mkdir -p $d
    if [ "${WALK_ONLY:-0}" = 1 ]; then
      cp $PKG/build/bh_engine.o $d/bh_engine.o
    else
      /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -w -ffp-contract=off "-DBHGPU_BUILD_INFO=\"digest=$DIGEST flags=$flags\"" $flags -c $PKG/csrc/bh_engine.hip -o $d/bh_engine.o || exit 1
    fi
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -w $flags -c $PKG/csrc/bh_walk_fast.hip -o $d/bh_walk_fast.o || exit 1
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $PKG/build/libbhgpu_$name.so $d/bh_engine.o $d/bh_walk_fast.o || exit 1
    echo "built $name ($flags)"
  ) &
  pids+=($!)
  if [ ${#pids[@]} -ge 6 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
