import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpu_nbody_simulation_amd as G
from gpu_nbody_simulation_amd import initial_conditions as IC
n = 1 << 20
for tm in (1e-3, 3.0):
    m, p, v = IC.plummer(n, 1, total_mass=tm)
    e = G.BarnesHutEngine(G.BhConfig(capacity=n, max_depth=21, precision=G.Precision.F32, reference_compat=False))
    e.upload(p, v, m)
    for s in range(4):
        e.compute_forces(); a = e.accelerations(); st = e.stats()
        bad = ~np.isfinite(a).all(axis=1)
        an = np.linalg.norm(a, axis=1)
        print("M", tm, "step", s, "n_nodes", st.n_nodes, "nonfinite acc", bad.sum(), "max|a| %.3e" % np.nanmax(an[~bad]) if (~bad).any() else "", "argmax", np.nanargmax(np.where(bad, -1, an)))
        if bad.any():
            i = np.where(bad)[0][:5]; print("  bad idx", i, a[i])
        e.step(1)
        pp, vv = e.download()
        print("   pos finite", np.isfinite(pp).all(), "vel finite", np.isfinite(vv).all(), "max|v| %.3e" % np.abs(vv[np.isfinite(vv)]).max(), "extent", pp[np.isfinite(pp).all(1)].min(0), pp[np.isfinite(pp).all(1)].max(0))
    e.close()
print("== bench flow: 3x step(1) + step(20)")
m, p, v = IC.plummer(n, 1)
e = G.BarnesHutEngine(G.BhConfig(capacity=n, max_depth=21, precision=G.Precision.F32, reference_compat=False))
e.upload(p, v, m)
for _ in range(3): e.step(1)
e.step(20)
pp, vv = e.download()
e.build_tree(); st = e.stats()
print(" finite", np.isfinite(pp).all(), np.isfinite(vv).all(), "n_nodes", st.n_nodes, "ms/step", st.last_step_ms, "walk", st.walk_ms, "extent", pp.min(0), pp.max(0), "max|v|", np.abs(vv).max())
same = len(pp) - len(np.unique(pp.astype(np.float32).view([('x','f4'),('y','f4')])))
print(" coincident bodies:", same)
