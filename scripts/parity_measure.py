"""Measured fp32 / mixed-precision parity of the walk against the oracle, body by body, split by whether the oracle
finds a BORDERLINE cell for the body (tests/parity_classes.py).  One JSON line per configuration; the tolerances of
tests/test_gpu_parity_classes.py and tests/test_gpu_configs.py are <= 2x these measurements.
    python scripts/parity_measure.py [c2 c2p c3 c3u c4 c5 t3um@1 ... ]      (name@seed: another seed of a configuration)"""
import dataclasses
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_classes as PC  # noqa: E402
from oracle import bh_oracle as O  # noqa: E402
import gpu_nbody_simulation_amd as G  # noqa: E402
from gpu_nbody_simulation_amd import initial_conditions as IC  # noqa: E402
from gpu_nbody_simulation_amd.engine import FLAG_WALK_STATS  # noqa: E402

CONFIGS = {
    # name: (kind, n, theta, precision, sample (0 = all bodies))
    "c2": ("uniform", 65536, 0.5, "f32", 0),
    "c2p": ("plummer", 65536, 0.5, "f32", 0),
    "c3": ("plummer", 1 << 20, 0.5, "f32", 0),
    "c3u": ("uniform", 1 << 20, 0.5, "f32", 0),
    "c4": ("plummer", 1 << 22, 0.5, "f32", 0),                 # round 4: ALL 4,194,304 bodies (round 3: 65,536)
    "c5": ("plummer", 1 << 24, 0.3, "mixed", 1 << 20),         # round 4: 1,048,576 sampled bodies (round 3: 65,536)
    "c5all": ("plummer", 1 << 24, 0.3, "mixed", 0),            # every one of the 16,777,216 bodies
    # tests/test_gpu_configs.py::test_theta_03_without_compat_against_the_uncapped_oracle (seed 7 there, 1 here)
    "t3u": ("uniform", 65536, 0.3, "f32", 0), "t3p": ("plummer", 65536, 0.3, "f32", 0),
    "t3um": ("uniform", 65536, 0.3, "mixed", 0), "t3pm": ("plummer", 65536, 0.3, "mixed", 0),
}


def measure(name):
    name, _, seed = name.partition("@")
    seed = int(seed or 1)
    kind, n, theta, prec, sample = CONFIGS[name]
    m, p, v = IC.make(kind, n, seed, quasi_static=True)
    if prec == "mixed":                                   # fp64 positions that are NOT fp32 values (config 5)
        p = p * (1.0 + 3e-9 * np.random.default_rng(seed).standard_normal(p.shape))
    m, p, v, s = PC.sample_first(m, p, v, sample or n)
    t0 = time.time()
    with G.BarnesHutEngine(G.BhConfig(capacity=n, theta=theta, max_depth=21, reference_compat=False, flags=FLAG_WALK_STATS,
                                      precision=G.Precision.MIXED if prec == "mixed" else G.Precision.F32)) as e:
        e.upload(p, v, m)
        e.compute_forces()
        a = e.accelerations()[:s]
        cnt = e.interaction_counts()[:s]
    t1 = time.time()
    tree = O.build_tree(p, m, 0)
    t2 = time.time()
    rep = PC.classify(a, cnt, m, p, theta, s, pos_rounded=(prec == "mixed"), tree=tree)
    t3 = time.time()
    print(json.dumps({"config": name, "seed": seed, "kind": kind, "n": n, "theta": theta, "precision": prec,
                      "seconds": {"gpu": round(t1 - t0, 1), "oracle_tree": round(t2 - t1, 1), "oracle_walk": round(t3 - t2, 1)},
                      **dataclasses.asdict(rep)}), flush=True)


if __name__ == "__main__":
    for name in (sys.argv[1:] or ["c2", "c2p", "c3", "c3u", "c4", "c5"]):
        measure(name)
