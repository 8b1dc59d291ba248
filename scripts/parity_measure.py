"""Measured fp32 / mixed-precision parity of the walk against the oracle, body by body, split by whether the
oracle finds a BORDERLINE cell for the body (one whose acceptance criterion fp32 arithmetic may decide the other
way, oracle/bh_oracle.c: bho_compute_forces_diag).  Prints one JSON line per configuration; the tolerances of
tests/test_gpu_parity_classes.py are <= 2x these measurements.
    python scripts/parity_measure.py [c2 c3 c4 c5 ...]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
    "c4": ("plummer", 1 << 22, 0.5, "f32", 65536),
    "c5": ("plummer", 1 << 24, 0.3, "mixed", 65536),
}


def sample_first(m, p, v, s):
    """Reorder the bodies so that the sample comes first: a quarter nearest to the centre of mass, a quarter
    farthest from it (core and halo), the rest a stride through the others."""
    n = len(m)
    r = np.linalg.norm(p - np.average(p, axis=0, weights=m), axis=1)
    order = np.argsort(r, kind="stable")
    core, halo, mid = order[:s // 4], order[n - s // 4:], order[s // 4:n - s // 4]
    stride = mid[::max(1, len(mid) // (s - 2 * (s // 4)))][:s - 2 * (s // 4)]
    chosen = np.concatenate([core, halo, stride])
    rest = np.setdiff1d(np.arange(n), chosen, assume_unique=False)
    perm = np.concatenate([chosen, rest])
    return m[perm], p[perm], v[perm], len(chosen)


def q(x, qs=(0.5, 0.99, 0.999, 1.0)):
    return [float(np.quantile(x, t)) if len(x) else None for t in qs]


def measure(name):
    kind, n, theta, prec, sample = CONFIGS[name]
    m, p, v = IC.make(kind, n, 1, quasi_static=True)
    if prec == "mixed":                                   # fp64 positions that are NOT fp32 values (config 5)
        rng = np.random.default_rng(11)
        p = p * (1.0 + rng.uniform(-2e-8, 2e-8, p.shape))
    s = n
    if sample:
        m, p, v, s = sample_first(m, p, v, sample)
    t0 = time.time()
    with G.BarnesHutEngine(G.BhConfig(capacity=n, theta=theta, max_depth=21, reference_compat=False, flags=FLAG_WALK_STATS,
                                      precision=G.Precision.MIXED if prec == "mixed" else G.Precision.F32)) as e:
        e.upload(p, v, m)
        e.compute_forces()
        a = e.accelerations()[:s]
        cnt = e.interaction_counts()[:s]
    t1 = time.time()
    tree = O.build_tree(p, m, 21)
    t2 = time.time()
    d = O.compute_forces_diag(tree, p, m, theta=theta, compat_self_skip=False, hi=s, pos_rounded=(prec == "mixed"))
    t3 = time.time()
    ao = d.forces[:s] / m[:s, None]
    # the reference divides by zero for exactly coincident bodies (inf * 0 = NaN, project.cu:651-658); the fp32 walk
    # lets such a pair contribute nothing (deviation iii): those bodies are counted, not compared
    ok = np.isfinite(ao).all(axis=1)
    n_nonfinite = int((~ok).sum())
    a, cnt, ao = a[ok], cnt[ok], ao[ok]
    d = O.WalkDiag(d.forces[:s][ok], d.counts[:s][ok], d.abs_sum[:s][ok], d.coord[:s][ok], d.flip[:s][ok])
    m = m[:s][ok]
    s = int(ok.sum())
    an = np.linalg.norm(ao, axis=1)
    err = np.linalg.norm(a - ao, axis=1)
    rel = err / an
    flip = d.flip[:s] / m[:s]
    clean = flip == 0
    ulp = 2.0 ** -24
    model = ulp * (16.0 * d.abs_sum[:s] + 4.0 * d.coord[:s]) / m[:s]            # a rounding-only forward bound
    out = {
        "config": name, "kind": kind, "n": n, "theta": theta, "precision": prec, "bodies_checked": int(s),
        "seconds": {"gpu": round(t1 - t0, 1), "oracle_tree": round(t2 - t1, 1), "oracle_walk": round(t3 - t2, 1)},
        "oracle_nonfinite_bodies": n_nonfinite,
        "clean_fraction": float(clean.mean()),
        "clean_counts_equal_fraction": float((cnt[clean] == d.counts[:s][clean]).mean()),
        "clean_rel_err_q50_q99_q999_max": q(rel[clean]),
        "clean_err_over_rounding_model_q50_q999_max": q((err / model)[clean], (0.5, 0.999, 1.0)),
        "borderline_bodies": int((~clean).sum()),
        "borderline_counts_equal_fraction": float((cnt[~clean] == d.counts[:s][~clean]).mean()) if (~clean).any() else None,
        "borderline_rel_err_q50_max": q(rel[~clean], (0.5, 1.0)),
        "borderline_err_minus_flip_budget_over_a_max": float(((err - flip) / an)[~clean].max()) if (~clean).any() else None,
        "all_rel_err_q50_q99_q999_max": q(rel),
        "clean_count_mismatches": int((cnt[clean] != d.counts[:s][clean]).sum()),
        "clean_mismatch_examples": [[int(i), int(cnt[i]), int(d.counts[i]), float(rel[i])] for i in np.flatnonzero(clean & (cnt != d.counts[:s]))[:8]],
        "interactions_gpu_vs_oracle": [int(cnt.sum()), int(d.counts[:s].sum())],
    }
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    for name in (sys.argv[1:] or ["c2", "c2p", "c3", "c3u", "c4", "c5"]):
        measure(name)
