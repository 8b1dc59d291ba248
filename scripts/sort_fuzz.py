"""Fuzz of the bucket sort against the LSD passes under violent dynamics (heavy bodies, no softening: ejections,
exploding root boxes, collapsed trees): same trajectories bit for bit, whatever happens to the keys.
  python scripts/sort_fuzz.py   (on a GPU box)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpu_nbody_simulation_amd as G  # noqa: E402

f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
bad = 0
for n, steps in ((5000, 60), (40000, 40), (300000, 25), (1200000, 12)):
    for seed in range(4):
        r = np.random.default_rng(100 * seed + n % 97)
        scale = 10.0 ** r.uniform(-6, 0)
        m = f32(scale * 10.0 ** r.uniform(-2, 1, n))
        p = f32(r.normal(0, 0.1, (n, 2)) if seed % 2 else r.uniform(-0.1, 0.1, (n, 2)))
        v = f32(r.normal(0, 1e-4, (n, 2)))
        out = []
        for mode in ("1", "0"):
            os.environ["BH_SORT_BUCKET"] = mode
            with G.BarnesHutEngine(G.BhConfig(capacity=n, precision=G.Precision.F32, max_depth=21,
                                              reference_compat=bool(seed & 2))) as e:
                e.upload(p, v, m)
                e.step(steps)
                st = e.stats()
                out.append(e.download() + (st.n_nodes, st.sort_spill_buckets))
        same = np.array_equal(out[0][0], out[1][0], equal_nan=True) and np.array_equal(out[0][1], out[1][1], equal_nan=True)
        bad += not same
        print(f"n={n} seed={seed} mass scale {scale:.1e}: nodes at the end {out[0][2]}, spills {out[0][3]}, "
              f"{'identical' if same else 'DIFFERENT'}", flush=True)
print("FAILED" if bad else "all identical")
sys.exit(1 if bad else 0)
