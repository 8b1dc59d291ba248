"""Worst CLEAN bodies of a parity_measure configuration, with everything the oracle knows about them."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_classes as PC
from oracle import bh_oracle as O
import gpu_nbody_simulation_amd as G
from gpu_nbody_simulation_amd import initial_conditions as IC
from gpu_nbody_simulation_amd.engine import FLAG_WALK_STATS
from parity_measure import CONFIGS

name = sys.argv[1]
kind, n, theta, prec, sample = CONFIGS[name]
m, p, v = IC.make(kind, n, 1, quasi_static=True)
if prec == "mixed":
    p = p * (1.0 + 3e-9 * np.random.default_rng(1).standard_normal(p.shape))
m, p, v, s = PC.sample_first(m, p, v, sample or n)
with G.BarnesHutEngine(G.BhConfig(capacity=n, theta=theta, max_depth=21, reference_compat=False, flags=FLAG_WALK_STATS,
                                  precision=G.Precision.MIXED if prec == "mixed" else G.Precision.F32)) as e:
    e.upload(p, v, m); e.compute_forces(); a = e.accelerations()[:s]; cnt = e.interaction_counts()[:s]
tree = O.build_tree(p, m, 0)
d = O.compute_forces_diag(tree, p, m, theta=theta, compat_self_skip=False, hi=s, pos_rounded=(prec == "mixed"))
ao = d.forces[:s] / m[:s, None]
an = np.linalg.norm(ao, axis=1); err = np.linalg.norm(a - ao, axis=1); rel = err / an
clean = d.flip[:s] == 0
model = 2.0 ** -24 * (16 * d.abs_sum[:s] + 4 * d.coord[:s]) / m[:s]
t21 = O.build_tree(p, m, 21)
# bodies that share a depth-cap cell in the depth-21 tree: particle == -1 leaves with mass of several bodies
order = np.argsort(-(err / model) * clean)[:12]
r = np.linalg.norm(p - np.average(p, axis=0, weights=m), axis=1)
# nearest neighbour distance of the worst bodies (brute force over all)
for i in order:
    dd = np.linalg.norm(p - p[i], axis=1); dd[i] = np.inf
    j = int(np.argmin(dd))
    print(json.dumps({"i": int(i), "rel": float(rel[i]), "err_over_model": float(err[i] / model[i]), "cnt_gpu": int(cnt[i]), "cnt_or": int(d.counts[i]),
                      "abs_over_a": float(d.abs_sum[i] / m[i] / an[i]), "coord_over_a": float(d.coord[i] / m[i] / an[i]), "radius": float(r[i]),
                      "nn_dist": float(dd[j]), "nn_dist_f32": float(np.linalg.norm(p[j].astype(np.float32) - p[i].astype(np.float32))),
                      "pos": [float(p[i][0]), float(p[i][1])]}))
print("mismatch clean:", int((clean & (cnt != d.counts[:s])).sum()))
for i in np.flatnonzero(clean & (cnt != d.counts[:s]))[:10]:
    dd = np.linalg.norm(p - p[i], axis=1); dd[i] = np.inf; j = int(np.argmin(dd))
    print(json.dumps({"i": int(i), "rel": float(rel[i]), "cnt_gpu": int(cnt[i]), "cnt_or": int(d.counts[i]), "nn_dist": float(dd[j]), "radius": float(r[i])}))
