"""step / build ms with 2, 4 or 8 keys per thread (tiles of 512 / 1,024 / 2,048: BH_BUILD_ITEMS) in the
sort / scan kernels of the build, over N: the data behind kSmallBuildBodies / kMediumBuildBodies in
bh_engine.hip.  python scripts/build_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpu_nbody_simulation_amd as G
from gpu_nbody_simulation_amd import initial_conditions as IC
for init in ("plummer",):
    for n in (131072, 262144, 393216, 524288, 786432, 1048576):
        m, p, v = IC.make(init, n, 1, quasi_static=True)
        out = []
        for val in ("8", "4", "2", "0"):
            os.environ["BH_BUILD_ITEMS"] = val
            with G.BarnesHutEngine(G.BhConfig(capacity=n, theta=0.5, max_depth=21, precision=G.Precision.F32, reference_compat=False)) as e:
                e.upload(p, v, m); e.step(5); e.sync()
                t0 = time.perf_counter(); e.step(100); e.sync()
                ms = (time.perf_counter() - t0) / 100 * 1e3
                st = e.stats()
            out.append("items=%s step %.3f build %.3f" % (val, ms, st.build_ms))
        print(init, n, "  |  ".join(out), flush=True)
