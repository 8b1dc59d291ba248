"""(the variants compared here are compiled only into an experiments build: `python -m gpu_nbody_simulation_amd.build --variant exp -DBHGPU_EXPERIMENTS`, then run with BHGPU_LIB_OPT_IN=1 BHGPU_LIB=gpu-nbody-simulation_amd/build/libbhgpu_exp.so)
build ms / step ms with the 3-launch radix pass vs the one-sweep pass (BH_SORT_ONESWEEP) over N."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpu_nbody_simulation_amd as G
from gpu_nbody_simulation_amd import initial_conditions as IC
for n in (1024, 16384, 65536, 131072, 262144, 524288, 1048576):
    m, p, v = IC.make("plummer", n, 1, quasi_static=True)
    out = []
    for os1 in ("0", "1"):
        os.environ["BH_SORT_ONESWEEP"] = os1
        with G.BarnesHutEngine(G.BhConfig(capacity=n, theta=0.5, max_depth=21, precision=G.Precision.F32, reference_compat=False)) as e:
            e.upload(p, v, m); e.step(5); e.sync()
            t0 = time.perf_counter(); e.step(50); e.sync()
            ms = (time.perf_counter() - t0) / 50 * 1e3
            st = e.stats()
        out.append("onesweep=%s step %.3f build %.3f" % (os1, ms, st.build_ms))
    print(n, "  ".join(out), flush=True)
