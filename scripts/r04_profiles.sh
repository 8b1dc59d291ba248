#!/bin/bash
# Round-4 profile set on the GPU box (one gpurun call per block below keeps each under the time limit):
#   scripts/r04_profiles.sh c3        headline: full bench line + kernel stats + PMC + calibration
#   scripts/r04_profiles.sh others    C4, C5, C2: kernel stats + FETCH / WRITE / SQ passes
#   scripts/r04_profiles.sh f64       BH_PRECISION_F64: kernel stats + PMC
#   scripts/r04_profiles.sh exact     BH_PRECISION_F64_EXACT: kernel stats + PMC
#   scripts/r04_profiles.sh emul      emulated ranks (N = 1M: W = 2, 4, 8; C4 and C5: W = 8) + kernel trace of one 8-rank step
#   scripts/r04_profiles.sh clock     in-kernel clock of the fp32 walk (experiments build: scripts/build_variants.sh exp:"-DBHGPU_EXPERIMENTS")
case "$1" in
c3) scripts/gpu_profile.sh r04_c3 > gpurun_out/r04_c3.log 2>&1; tail -2 gpurun_out/r04_c3.log ;;
others)
  scripts/gpu_profile_config.sh r04_c4 --n-bodies 4194304 --steps 10 --warmup 2 > gpurun_out/r04_c4.log 2>&1; tail -1 gpurun_out/r04_c4.log
  scripts/gpu_profile_config.sh r04_c5 --n-bodies 16777216 --theta 0.3 --precision mixed --steps 5 --warmup 1 > gpurun_out/r04_c5.log 2>&1; tail -1 gpurun_out/r04_c5.log
  TRACE_STEPS=200 scripts/gpu_profile_config.sh r04_c2 --n-bodies 65536 --init uniform --steps 1000 --warmup 20 > gpurun_out/r04_c2.log 2>&1; tail -1 gpurun_out/r04_c2.log ;;
f64) bash scripts/f64_pmc.sh r04_f64 > gpurun_out/r04_f64.log 2>&1; tail -3 gpurun_out/r04_f64.log ;;
exact) RUN="python3 scripts/exact_run.py 1048576" bash scripts/f64_pmc.sh r04_exact > gpurun_out/r04_exact.log 2>&1; tail -3 gpurun_out/r04_exact.log ;;
emul)
  mkdir -p gpurun_out/r04_emul
  python scripts/let_emulate.py --n 1048576 --worlds 2,4,8 > gpurun_out/r04_emul/emulation_1m.txt 2>&1; tail -3 gpurun_out/r04_emul/emulation_1m.txt
  python scripts/let_emulate.py --n 4194304 --worlds 8 > gpurun_out/r04_emul/emulation_4m.txt 2>&1; tail -1 gpurun_out/r04_emul/emulation_4m.txt
  python scripts/let_emulate.py --n 16777216 --worlds 8 --theta 0.3 --precision mixed --reps 5 > gpurun_out/r04_emul/emulation_16m.txt 2>&1; tail -1 gpurun_out/r04_emul/emulation_16m.txt
  cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_emul/trace -- python3 scripts/let_emulate.py --n 1048576 --worlds 8 --reps 10 > gpurun_out/r04_emul/trace.log 2>&1; tail -1 gpurun_out/r04_emul/trace.log ;;
clock)
  mkdir -p gpurun_out/r04_clock
  export BHGPU_LIB_OPT_IN=1 BHGPU_LIB=$PWD/gpu-nbody-simulation_amd/build/libbhgpu_exp.so BH_WALK_TIMELINE=/tmp/tl.bin
  python scripts/walk_timeline.py 1048576 plummer 0.5 f32 5000 --json gpurun_out/r04_clock/walk_clock_c3.json > gpurun_out/r04_clock/timeline_c3.txt 2>&1; head -3 gpurun_out/r04_clock/timeline_c3.txt
  python scripts/walk_timeline.py 16777216 plummer 0.3 mixed 200 --json gpurun_out/r04_clock/walk_clock_c5.json > gpurun_out/r04_clock/timeline_c5.txt 2>&1; head -3 gpurun_out/r04_clock/timeline_c5.txt ;;
*) echo "usage: $0 c3|others|f64|exact|emul|clock"; exit 2 ;;
esac
