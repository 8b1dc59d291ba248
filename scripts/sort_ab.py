"""(the variants compared here are compiled only into an experiments build: `python -m gpu_nbody_simulation_amd.build --variant exp -DBHGPU_EXPERIMENTS`, then run with BHGPU_LIB_OPT_IN=1 BHGPU_LIB=gpu-nbody-simulation_amd/build/libbhgpu_exp.so)
build / step ms with the wave-private-ranking scatter (BH_SORT_WAVE_RANK=1, default) and the
workgroup-ranked one (0).  python scripts/sort_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpu_nbody_simulation_amd as G
from gpu_nbody_simulation_amd import initial_conditions as IC
for n in (65536, 262144, 1048576, 4194304):
    m, p, v = IC.make("plummer", n, 1, quasi_static=True)
    out = []
    for val in ("0", "1"):
        os.environ["BH_SORT_WAVE_RANK"] = val
        with G.BarnesHutEngine(G.BhConfig(capacity=n, theta=0.5, max_depth=21, precision=G.Precision.F32, reference_compat=False)) as e:
            e.upload(p, v, m); e.step(5); e.sync()
            t0 = time.perf_counter(); e.step(50); e.sync()
            ms = (time.perf_counter() - t0) / 50 * 1e3
            st = e.stats()
            pos, _ = e.download()
        out.append((ms, st.build_ms, pos))
    same = (out[0][2] == out[1][2]).all()
    print(f"N={n}: workgroup-ranked step {out[0][0]:.3f} build {out[0][1]:.3f} | wave-ranked step {out[1][0]:.3f} build {out[1][1]:.3f} | same trajectories: {same}", flush=True)
