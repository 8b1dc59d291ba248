"""Run K steps of one workload on cuda:0 and print the step / build / sort times and the bucket-sort spills
(for rocprofv3 --kernel-trace --stats -- python3 scripts/run_steps.py ...).  Development aid, not a bench."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpu_nbody_simulation_amd as G  # noqa: E402
from gpu_nbody_simulation_amd import initial_conditions as IC  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1 << 20)
ap.add_argument("--init", default="plummer")
ap.add_argument("--drift", type=float, default=0.0)
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--theta", type=float, default=0.5)
a = ap.parse_args()
m, p, v = IC.make(a.init, a.n, 1, quasi_static=True, drift_cells=a.drift)
with G.BarnesHutEngine(G.BhConfig(capacity=a.n, theta=a.theta, max_depth=21, precision=G.Precision.F32,
                                  reference_compat=False)) as e:
    e.upload(p, v, m)
    e.step(a.warmup)
    e.sync()
    t0 = time.perf_counter()
    e.step(a.steps)
    e.sync()
    dt = (time.perf_counter() - t0) / a.steps * 1e3
    st = e.stats()
print(json.dumps({"n": a.n, "init": a.init, "drift": a.drift, "ms_per_step": round(dt, 4), "build_ms": round(st.build_ms, 4),
                  "walk_ms": round(st.walk_ms, 4), "keys_ms": round(st.keys_ms, 4), "sort_ms": round(st.sort_ms, 4),
                  "scan_ms": round(st.scan_ms, 4), "nodes_ms": round(st.nodes_ms, 4), "sort_spill_buckets": st.sort_spill_buckets, "sort_rerun_buckets": st.sort_rerun_buckets}))
