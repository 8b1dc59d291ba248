#!/bin/bash
# One non-headline configuration on the GPU box: rocprofv3 kernel stats + the FETCH / WRITE PMC passes of the same
# command.  usage: scripts/gpu_profile_config.sh <tag> <bench args...>
set -e
TAG=$1; shift
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
B="--no-cpu-baseline --no-secondary"
python3 bench.py $B "$@" > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $B "$@" ${TRACE_STEPS:+--steps $TRACE_STEPS} > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 bench.py $B "$@" --steps 3 --warmup 1 > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc4 -- python3 bench.py $B "$@" --steps 3 --warmup 1 > $OUT/pmc4.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $OUT/pmc1 -- python3 bench.py $B "$@" --steps 3 --warmup 1 > $OUT/pmc1.log 2>&1
find $OUT -name "*.csv" | wc -l
