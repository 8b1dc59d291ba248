#!/bin/bash
# Per-rank device times of the two multi-GPU decompositions emulated on one GPU
mkdir -p gpurun_out
python scripts/let_emulate.py --n 1048576 --worlds 2,4,8 > gpurun_out/emul_1m.txt 2>&1; tail -4 gpurun_out/emul_1m.txt
python scripts/let_emulate.py --n 4194304 --worlds 8 > gpurun_out/emul_4m.txt 2>&1; tail -2 gpurun_out/emul_4m.txt
python scripts/let_emulate.py --n 16777216 --worlds 8 --theta 0.3 --precision mixed --reps 5 > gpurun_out/emul_16m.txt 2>&1; tail -2 gpurun_out/emul_16m.txt
python scripts/let_emulate.py --n 1048576 --worlds 8 --init uniform > gpurun_out/emul_1m_u.txt 2>&1; tail -2 gpurun_out/emul_1m_u.txt
