"""BH_PRECISION_F64 at N = 1M Plummer: ms/step, walk, build (scripts/lib_ab.py-style A/B of library variants)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpu_nbody_simulation_amd as G
from gpu_nbody_simulation_amd import initial_conditions as IC
from oracle import bh_oracle as O
from gpu_nbody_simulation_amd.engine import FLAG_WALK_STATS
n = 1 << 20
m, p, v = IC.make("plummer", n, 1, quasi_static=True)
with G.BarnesHutEngine(G.BhConfig(capacity=n, theta=0.5, max_depth=21, precision=G.Precision.F64, reference_compat=True)) as e:
    e.upload(p, v, m); e.step(3); e.sync()
    best = None
    for _ in range(3):
        t0 = time.perf_counter(); e.step(10); e.sync(); dt = (time.perf_counter() - t0) / 10 * 1e3
        st = e.stats()
        best = min(best, (dt, st.walk_ms, st.build_ms)) if best else (dt, st.walk_ms, st.build_ms)
# accuracy on a slice
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_project_40960.npz"))
with G.BarnesHutEngine(G.BhConfig(capacity=40960, precision=G.Precision.F64, flags=FLAG_WALK_STATS)) as e:
    e.upload(g["pos"], g["vel"], g["mass"]); f = e.compute_forces(); cnt = e.interaction_counts()
d = O.compute_forces_diag(O.build_tree(g["pos"], g["mass"], 10), g["pos"], g["mass"], compat_self_skip=True)
rel = np.linalg.norm(f - g["forces_0"], axis=1) / np.linalg.norm(g["forces_0"], axis=1)
print(json.dumps({"lib": os.path.basename(os.environ.get("BHGPU_LIB", "libbhgpu.so")), "ms_per_step": round(best[0], 4), "walk_ms": round(best[1], 4),
                  "build_ms": round(best[2], 4), "max_rel_err_vs_golden_40960": float(rel.max()), "counts_equal": bool(np.array_equal(cnt, d.counts))}))
