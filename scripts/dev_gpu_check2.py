"""fp32 bucket mode vs uncapped oracle; timing of the step at several N."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import bh_oracle as O
import gpu_nbody_simulation_amd as G
from gpu_nbody_simulation_amd import initial_conditions as IC
from gpu_nbody_simulation_amd.engine import FLAG_WALK_STATS, FLAG_LDS_STACK

for kind, n, md in (("plummer", 65536, 12), ("plummer", 65536, 21), ("uniform", 65536, 21)):
    m, p, v = IC.make(kind, n, 1)
    t = O.build_tree(p, m, 0)
    ao = O.compute_forces(t, p, m, compat_self_skip=False) / m[:, None]
    e = G.BarnesHutEngine(G.BhConfig(capacity=n, max_depth=md, precision=G.Precision.F32, reference_compat=False, flags=FLAG_WALK_STATS))
    e.upload(p, v, m); e.compute_forces(); a = e.accelerations(); st = e.stats()
    rel = np.linalg.norm(a - ao, axis=1) / np.linalg.norm(ao, axis=1)
    print(kind, n, "md", md, "rel err median %.2e p99 %.2e p99.9 %.2e max %.2e" % (np.median(rel), np.quantile(rel, .99), np.quantile(rel, .999), rel.max()),
          "nodes", st.n_nodes, len(t), "U64 %.1f inter/body %.1f" % (st.wave_nodes / n, st.interactions / n), flush=True)
    e.close()

for kind, n, md in (("uniform", 65536, 16), ("uniform", 1 << 20, 21), ("plummer", 1 << 20, 21), ("plummer", 1 << 22, 21)):
    m, p, v = IC.make(kind, n, 1)
    for flags in (0, FLAG_LDS_STACK):
        e = G.BarnesHutEngine(G.BhConfig(capacity=n, max_depth=md, precision=G.Precision.F32, reference_compat=False, flags=flags))
        e.upload(p, v, m)
        e.step(3); e.sync()
        t0 = time.perf_counter(); e.step(10); e.sync(); t1 = time.perf_counter()
        st = e.stats()
        print(kind, n, "md", md, "flags", flags, "ms/step %.3f (events %.3f) build %.3f walk %.3f  -> %.1f M body-steps/s" % ((t1 - t0) * 100, st.last_step_ms, st.build_ms, st.walk_ms, n / ((t1 - t0) / 10) / 1e6), flush=True)
        e.close()
