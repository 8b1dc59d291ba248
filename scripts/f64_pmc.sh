#!/bin/bash
# BH_PRECISION_F64 at N = 1M on the GPU box: kernel stats + the PMC passes of walk_f64_kernel (SQ issue counters, FETCH, WRITE).
# usage: [RUN=...] scripts/f64_pmc.sh <tag>   -> gpurun_out/<tag>/{trace,pmc1..4}; scripts/summarize_profile.py <that> profiles/<name> --tag F64
TAG=$1; OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
R="${RUN:-python3 scripts/f64_run.py 1048576}"        # RUN="python3 scripts/exact_run.py 1048576": the bit-exact mode
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $R 20 > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/pmc1 -- $R 3 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc2 -- $R 3 > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- $R 3 > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc4 -- $R 3 > $OUT/pmc4.log 2>&1
tail -2 $OUT/trace.log
find $OUT -name "*.csv" | wc -l
