#!/bin/bash
# Round-3 profile set on the GPU box: headline (C3) full bench + kernel stats + PMC, then C4, C5 and C2 (kernel stats + traffic),
# the fast-fp64 mode's kernel stats, then the kernel trace of one emulated 8-rank step.
scripts/gpu_profile.sh r03_c3 > gpurun_out/r03_c3.log 2>&1; tail -2 gpurun_out/r03_c3.log
scripts/gpu_profile_config.sh r03_c4 --n-bodies 4194304 --steps 10 --warmup 2 > gpurun_out/r03_c4.log 2>&1; tail -1 gpurun_out/r03_c4.log
scripts/gpu_profile_config.sh r03_c5 --n-bodies 16777216 --theta 0.3 --precision mixed --steps 5 --warmup 1 > gpurun_out/r03_c5.log 2>&1; tail -1 gpurun_out/r03_c5.log
TRACE_STEPS=200 scripts/gpu_profile_config.sh r03_c2 --n-bodies 65536 --init uniform --steps 1000 --warmup 20 > gpurun_out/r03_c2.log 2>&1; tail -1 gpurun_out/r03_c2.log
bash scripts/kstats.sh r03_f64 scripts/f64_run.py 1048576 20 > gpurun_out/r03_f64.log 2>&1; tail -3 gpurun_out/r03_f64.log
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_emul8/trace -- python3 scripts/let_emulate.py --n 1048576 --worlds 8 --reps 10 > gpurun_out/r03_emul8.log 2>&1; tail -2 gpurun_out/r03_emul8.log
