"""Body-by-body parity of the fp32 / mixed-precision walk with the oracle, split by what can explain a difference
(VERDICT r2 #2).  Helper of tests/test_gpu_parity_classes.py and tests/test_gpu_configs.py -- not a test file.

The oracle's diagnostic walk (oracle/bh_oracle.c: bho_compute_forces_diag; forces bit-identical to the pinned walk
of project.cu:593-675) tells for every body whether ANY subdivided cell on its walk has an acceptance criterion
`size / d < theta` (project.cu:643) that fp32 arithmetic could decide the other way -- |d - size/theta| within the
uncertainty 2^-23 * ((|comx| + |comy|) / d + 4) * d that fp32 coordinates leave about d -- and if so, by how much
such flips can change the force (the cells' multipole errors, summed: the body's FLIP BUDGET).

  CLEAN bodies (no such cell; > 99.5 % of all): the device must accept EXACTLY the oracle's node set -- checked
  through the per-body count of accepted force evaluations (bh_get_interaction_counts), equal for 100 % of
  them -- so what is left is rounding, and it is bounded two ways:
    * relative to |a|: median / 99.9 % / max <= the stated tolerances (<= 2 x the measured values, DESIGN.md
      section 7).  SURVEY 8(c) hoped for 1e-5 at 99.9 %; measured on clean bodies it is 2e-5 .. 1.1e-4 -- not
      criterion flips but cancellation: a body near the centre of a cluster feels ~400 pulls that nearly cancel,
      and fp32 rounds each of them relative to ITS size;
    * relative to a forward rounding model, 2^-24 * ((1.5 * sqrt(n_i) + 8) * sum |a_j| + 4 * sum |a_j| * (|c_j| +
      |p|) / d_j over the accepted cells), n_i = the body's interaction count (a sum of n terms in fp32 drifts by
      ~sqrt(n) ulps of its partial sums): <= MODEL_MAX for EVERY clean body -- the statement with no exceptions.
  BORDERLINE bodies: |a_gpu - a_oracle| <= flip budget + the clean bound; their counts may differ.
The oracle walks the UNCAPPED tree (main_approach_2.cpp's) with ONE documented deviation switched on: a subdivided
cell at the device's depth cap is summed body by body for every body that reaches it (cap_depth: the device's
depth-cap bucket, DESIGN.md section 4 deviation iv -- more exact than the reference's multipole there); how much that
changes is reported per configuration (cap_affected, cap_max) and is nil below ~4M bodies.
Bodies for which the reference itself yields no finite force (exactly coincident pairs: inf * 0, project.cu:651-658)
are counted and skipped: the fp32 walk lets such a pair contribute nothing (deviation iii)."""
from dataclasses import dataclass

import numpy as np

from oracle import bh_oracle as O

MODEL_MAX = 1.0          # every clean body: error <= the forward rounding model (measured maxima in DESIGN.md section 7)


@dataclass
class ClassReport:
    bodies: int
    nonfinite: int
    clean_fraction: float
    clean_q50: float
    clean_q999: float
    clean_max: float
    clean_model_max: float
    clean_count_mismatches: int
    borderline: int
    borderline_excess_max: float      # max over borderline bodies of (err - flip budget) / |a|
    all_max: float
    cap_affected: int                 # bodies that reach a multi-body depth-cap cell (bucket semantics applied)
    cap_max: float                    # largest relative change that made against the plain uncapped walk


def sample_first(m, p, v, s):
    """Reorder the bodies so that the `s` sampled ones come first: a quarter nearest to the centre of mass, a
    quarter farthest from it (core and halo), the rest a stride through the others.  Returns (m, p, v, s)."""
    n = len(m)
    if s >= n:
        return m, p, v, n
    r = np.linalg.norm(p - np.average(p, axis=0, weights=m), axis=1)
    order = np.argsort(r, kind="stable")
    q = s // 4
    core, halo, mid = order[:q], order[n - q:], order[q:n - q]
    stride = mid[::max(1, len(mid) // (s - 2 * q))][:s - 2 * q]
    chosen = np.concatenate([core, halo, stride])
    mask = np.ones(n, dtype=bool)
    mask[chosen] = False
    perm = np.concatenate([chosen, np.flatnonzero(mask)])
    return m[perm], p[perm], v[perm], len(chosen)


def classify(a_gpu, counts_gpu, m, p, theta, s, pos_rounded=False, tree=None, cap_depth=21) -> ClassReport:
    """a_gpu, counts_gpu: accelerations and interaction counts of bodies [0, s) from the device; the oracle walks the
    same bodies through the UNCAPPED tree (main_approach_2.cpp's; compat off sums a depth-cap cell body by body)."""
    if tree is None:
        tree = O.build_tree(p, m, 0)
    d = O.compute_forces_diag(tree, p, m, theta=theta, compat_self_skip=False, hi=s, pos_rounded=pos_rounded,
                              cap_depth=cap_depth)
    ms = m[:s]
    ao = d.forces[:s] / ms[:, None]
    ok = np.isfinite(ao).all(axis=1)
    an = np.linalg.norm(ao[ok], axis=1)
    err = np.linalg.norm(a_gpu[:s][ok] - ao[ok], axis=1)
    rel = err / an
    flip = d.flip[:s][ok] / ms[ok]
    clean = flip == 0
    model = 2.0 ** -24 * ((1.5 * np.sqrt(d.counts[:s][ok]) + 8.0) * d.abs_sum[:s][ok] + 4.0 * d.coord[:s][ok]) / ms[ok]
    mism = int((counts_gpu[:s][ok][clean] != d.counts[:s][ok][clean]).sum())
    b = ~clean
    return ClassReport(
        bodies=int(ok.sum()), nonfinite=int((~ok).sum()), clean_fraction=float(clean.mean()),
        clean_q50=float(np.median(rel[clean])), clean_q999=float(np.quantile(rel[clean], 0.999)),
        clean_max=float(rel[clean].max()), clean_model_max=float((err / model)[clean].max()),
        clean_count_mismatches=mism, borderline=int(b.sum()),
        borderline_excess_max=float(((err - flip) / an)[b].max()) if b.any() else 0.0, all_max=float(rel.max()),
        cap_affected=int((d.cap[:s][ok] > 0).sum()), cap_max=float((d.cap[:s][ok] / ms[ok] / an).max()))


def check(rep: ClassReport, tol, min_clean=0.995):
    """tol = (median, 99.9 %, max) of the clean bodies' relative error."""
    q50, q999, mx = tol
    assert rep.clean_fraction >= min_clean, rep
    assert rep.clean_count_mismatches == 0, rep               # the same accepted node set, body by body
    assert rep.clean_q50 <= q50 and rep.clean_q999 <= q999 and rep.clean_max <= mx, rep
    assert rep.clean_model_max <= MODEL_MAX, rep               # no clean body beyond the forward rounding bound
    assert rep.borderline_excess_max <= mx, rep                # borderline bodies: the flip budget explains the rest
    assert rep.nonfinite <= 1e-3 * max(rep.bodies, 1), rep
    # the oracle applies the ONE documented deviation of the device tree itself (a multi-body cell at depth 21 is summed
    # body by body instead of being subdivided further), so that path must stay the rare, bounded thing it is: measured, 9 of
    # 4,194,304 bodies at C4 (largest change of a body's force 0.11 of it), 1,502 of 16,777,216 at C5 (0.082), none up to N = 1M
    assert rep.cap_affected <= 3e-4 * max(rep.bodies, 1) and rep.cap_max <= 0.25, rep
