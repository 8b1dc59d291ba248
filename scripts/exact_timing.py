"""ms/step of the bit-exact fp64 mode (build, node kernel, walk) at a few sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpu_nbody_simulation_amd as G
from gpu_nbody_simulation_amd import initial_conditions as IC
for n, md, kind in ((1 << 20, 21, "plummer"), (1024, 10, "uniform"), (65536, 10, "uniform")):
    m, p, v = IC.make(kind, n, 1, quasi_static=True)
    with G.BarnesHutEngine(G.BhConfig(capacity=n, max_depth=md)) as e:
        e.upload(p, v, m); e.step(2); e.sync()
        k = 5 if n > 100000 else 100
        t0 = time.perf_counter(); e.step(k); e.sync(); dt = (time.perf_counter() - t0) / k * 1e3
        st = e.stats()
    print("exact n=%d cap %d: %.4f ms/step build %.4f (nodes %.4f) walk %.4f" % (n, md, dt, st.build_ms, st.nodes_ms, st.walk_ms))
