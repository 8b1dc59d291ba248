"""step ms with the periodic physical re-ordering of the state (BH_REORDER_EVERY=16, default) and
without (0).  python scripts/reorder_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpu_nbody_simulation_amd as G
from gpu_nbody_simulation_amd import initial_conditions as IC
for n in (65536, 1 << 20, 1 << 22, 1 << 24):
    m, p, v = IC.make("plummer", n, 1, quasi_static=True)
    out = []
    for val in ("0", "16", "1"):
        os.environ["BH_REORDER_EVERY"] = val
        with G.BarnesHutEngine(G.BhConfig(capacity=n, theta=0.5, max_depth=21, precision=G.Precision.F32, reference_compat=False)) as e:
            e.upload(p, v, m); e.step(3); e.sync()
            k = 32 if n <= (1 << 22) else 16
            t0 = time.perf_counter(); e.step(k); e.sync()
            ms = (time.perf_counter() - t0) / k * 1e3
            st = e.stats()
        out.append("every=%s step %.3f build %.3f walk %.3f" % (val, ms, st.build_ms, st.walk_ms))
    print(f"N={n}: " + "  |  ".join(out), flush=True)
