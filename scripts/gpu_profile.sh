#!/bin/bash
# Runs on the GPU box (via gpurun): bench line + rocprofv3 kernel stats + PMC passes (+ calibration).
# usage: scripts/gpu_profile.sh <tag> [bench args...]
set -e
TAG=$1; shift
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --no-secondary "$@" > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/pmc1 -- python3 bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 "$@" > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc2 -- python3 bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 "$@" > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 "$@" > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc4 -- python3 bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 "$@" > $OUT/pmc4.log 2>&1
if [ -f scripts/calib/fetch_calib.hip ]; then
  hipcc -O3 --offload-arch=gfx950 scripts/calib/fetch_calib.hip -o /tmp/fetch_calib 2> $OUT/calib_build.log
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/calib -- /tmp/fetch_calib > $OUT/calib.log 2>&1
fi
find $OUT -name "*.csv" | wc -l
