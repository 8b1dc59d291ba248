"""fp32 parity where there is no excuse: BASELINE configs 2 and 3 walked body by body through the
oracle -- ALL 65,536 and ALL 1,048,576 bodies -- and split by what can explain a difference (tests/parity_classes.py):
bodies whose walk meets no borderline acceptance criterion (> 99.5 %) must accept EXACTLY the oracle's node set
(per-body interaction counts equal, 100 % of them) and differ from it by rounding only; the rest may differ by their
flip budget, the summed multipole error of their borderline cells.  Configs 4 and 5: tests/test_gpu_configs.py.

Tolerances (median, 99.9 %, max of the CLEAN bodies' relative acceleration error) are <= 2x the values measured by
scripts/parity_measure.py on MI355X (DESIGN.md section 7 has the table)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import parity_classes as PC  # noqa: E402
import gpu_nbody_simulation_amd as G  # noqa: E402
from gpu_nbody_simulation_amd import initial_conditions as IC  # noqa: E402
from gpu_nbody_simulation_amd.engine import FLAG_WALK_NO_SPLIT, FLAG_WALK_PORTABLE, FLAG_WALK_STATS  # noqa: E402

# config: (kind, n, theta, (median, 99.9 %, max) of the clean bodies' relative error); measured on MI355X
# (scripts/parity_measure.py, profiles/r03_final/parity_classes.txt): C2 3.7e-7 / 3.2e-5 / 5.0e-4, C2-plummer
# 1.9e-7 / 1.8e-5 / 1.6e-4, C3 5.7e-7 / 6.6e-5 / 1.4e-3, C3-uniform 1.3e-6 / 1.1e-4 / 3.5e-3
CASES = {
    "C2": ("uniform", 65536, 0.5, (7.5e-7, 6.4e-5, 1.0e-3)),
    "C2-plummer": ("plummer", 65536, 0.5, (3.7e-7, 3.7e-5, 3.3e-4)),
    "C3": ("plummer", 1 << 20, 0.5, (1.14e-6, 1.3e-4, 2.9e-3)),
    "C3-uniform": ("uniform", 1 << 20, 0.5, (2.7e-6, 2.2e-4, 6.9e-3)),
}


@pytest.mark.parametrize("case", list(CASES))
def test_every_body_against_the_oracle_by_class(case):
    kind, n, theta, tol = CASES[case]
    m, p, v = IC.make(kind, n, 1, quasi_static=True)
    with G.BarnesHutEngine(G.BhConfig(capacity=n, theta=theta, max_depth=21, precision=G.Precision.F32,
                                      reference_compat=False, flags=FLAG_WALK_STATS)) as e:
        e.upload(p, v, m)
        e.compute_forces()
        a = e.accelerations()
        cnt = e.interaction_counts()
        st = e.stats()
    assert int(cnt.sum()) == st.interactions                 # the per-body counts are the kernel's own counter, split
    rep = PC.classify(a, cnt, m, p, theta, n)
    PC.check(rep, tol)
    assert rep.cap_affected == 0                             # (no multi-body depth-cap cell is reached below ~4M bodies)
    # SURVEY 8(c) hoped for <= 1e-3 on every body nothing excuses.  Measured, the clean bodies' maximum is 5e-4 at C2
    # and 1.4e-3 / 3.5e-3 at C3: a few bodies at the centre of the cloud, where ~450 pulls cancel to 1/600 of their
    # summed magnitude -- every one of them inside the forward rounding bound (PC.MODEL_MAX, measured <= 0.65), which
    # is the bound that has no exceptions; against the SUM of the pulls no clean body is off by more than 3e-6.


def test_counting_walk_and_product_walk_are_the_same_walk():
    """The counts come from the counting variant of the kernel (C++ loop); the product runs the hand-scheduled loop.
    Same abstract machine: accelerations bitwise equal, so the counts describe the product's decisions."""
    n = 200000
    m, p, v = IC.make("plummer", n, 2, quasi_static=True)
    acc = []
    for flags in (FLAG_WALK_NO_SPLIT, FLAG_WALK_NO_SPLIT | FLAG_WALK_STATS, FLAG_WALK_NO_SPLIT | FLAG_WALK_PORTABLE):
        with G.BarnesHutEngine(G.BhConfig(capacity=n, max_depth=21, precision=G.Precision.F32, reference_compat=False,
                                          flags=flags)) as e:
            e.upload(p, v, m)
            e.compute_forces()
            acc.append(e.accelerations())
    assert np.array_equal(acc[0], acc[1]) and np.array_equal(acc[0], acc[2])


def test_random_systems_by_class():
    """40 seeded random systems in fp32 (64 to 40,000 bodies -- the one-wave walk, and the level-synchronous walk over 2, 4 and 8
    waves per group, whichever the launch size picks; 1 to 4 clusters of widths 1e-3 to 0.3; masses over four decades;
    theta 0.2 to 1.2): every body whose walk meets no borderline criterion accepts EXACTLY the oracle's node set (per-body
    counts), stays inside the forward rounding bound, and the borderline bodies inside their flip budget."""
    rng = np.random.default_rng(314)
    for case in range(40):
        n = int(2 ** rng.uniform(6, 15.3))
        theta = float(rng.uniform(0.2, 1.2))
        centres = rng.uniform(-1, 1, (int(rng.integers(1, 5)), 2))
        p = (centres[rng.integers(0, len(centres), n)] + rng.normal(0, 10.0 ** rng.uniform(-3, -0.5), (n, 2))).astype(np.float32).astype(np.float64)
        m = (10.0 ** rng.uniform(-2, 2, n)).astype(np.float32).astype(np.float64)
        v = np.zeros((n, 2))
        with G.BarnesHutEngine(G.BhConfig(capacity=n, theta=theta, max_depth=21, precision=G.Precision.F32,
                                          reference_compat=False, flags=FLAG_WALK_STATS)) as e:
            e.upload(p, v, m)
            e.compute_forces()
            a = e.accelerations()
            cnt = e.interaction_counts()
            st = e.stats()
        assert int(cnt.sum()) == st.interactions, (case, n, theta)
        rep = PC.classify(a, cnt, m, p, theta, n)
        assert rep.clean_count_mismatches == 0, (case, n, theta, rep)
        assert rep.clean_model_max <= PC.MODEL_MAX, (case, n, theta, rep)
        assert rep.borderline_excess_max <= 5e-2 and rep.nonfinite == 0, (case, n, theta, rep)
        # (no fixed relative tolerance: a cluster 1e-3 wide at distance 1 from the origin keeps three digits fewer of its
        #  coordinate differences in fp32 than one at the origin -- the rounding bound above prices exactly that)
        assert rep.clean_fraction >= 0.9, (case, n, theta, rep)
