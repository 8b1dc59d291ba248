"""GPU parity, BH_PRECISION_F64 (the reference's arithmetic type at throughput, csrc/bh_walk_f64.hpp).

The tree is the exact mode's -- checked bitwise against the reference's golden tree -- and the walk may visit nodes
in another order, decides acceptance on d^2 and takes 1/d from v_rsq_f64 + one Newton step, so the stated tolerance is on the FORCES:
    per body |F_gpu - F_ref| <= 1e-12 * |F_ref|      (reference = golden forces of project.cu's CPU path / the oracle)
    per body: the same number of accepted force evaluations as the oracle (the same acceptance decisions)
and on short trajectories <= 1e-11 x box width (the integration is the reference's kick-drift in fp64; the
reference's dynamics amplify a 1e-15 difference by the step, so long runs are compared statistically only)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import bh_oracle as O  # noqa: E402
import gpu_nbody_simulation_amd as G  # noqa: E402
from gpu_nbody_simulation_amd import initial_conditions as IC  # noqa: E402
from gpu_nbody_simulation_amd.engine import FLAG_WALK_PORTABLE, FLAG_WALK_STATS  # noqa: E402

TOL = 1e-12


def rel(f, ref):
    return np.linalg.norm(f - ref, axis=1) / np.linalg.norm(ref, axis=1)


def engine(n, **kw):
    kw.setdefault("precision", G.Precision.F64)
    return G.BarnesHutEngine(G.BhConfig(capacity=n, **kw))


def test_reference_forces_1024_and_the_reference_tree(gold, init1024):
    """BASELINE config[0]'s bodies, the reference's depth cap and self skip: the tree bitwise the golden tree, the
    forces within 1e-12 of the reference's own (golden), the walk's counters the oracle's."""
    m, p, v = init1024
    g = gold("ref_project_1024")
    with engine(1024, flags=FLAG_WALK_STATS) as e:
        e.upload(p, v, m)
        f = e.compute_forces()
        st = e.stats()
        nodes, depth = e.export_tree()
        cnt = e.interaction_counts()
    rn, rd = O.canonical_tree(g["tree_0"])
    assert np.array_equal(depth, rd)
    for fld in ("comx", "comy", "mass", "xmin", "xmax", "ymin", "ymax", "particle"):
        assert np.array_equal(nodes[fld], rn[fld]), fld
    assert rel(f, g["forces_0"]).max() <= TOL
    assert (st.visits, st.interactions) == (150509, 104117)                     # the oracle's counts (test_gpu_exact.py)
    d = O.compute_forces_diag(O.build_tree(p, m, 10), p, m, compat_self_skip=True)
    assert np.array_equal(cnt, d.counts)


@pytest.mark.parametrize("name,md,compat", [("ref_project_40960", 10, True), ("ref_project_40960", 32, False),
                                            ("ref_project_4096_grid", 10, True)])
def test_forces_against_the_oracle_body_by_body(gold, name, md, compat):
    g = gold(name)
    m, p, v = g["mass"], g["pos"], g["vel"]
    n = len(m)
    with engine(n, max_depth=md, reference_compat=compat, flags=FLAG_WALK_STATS) as e:
        e.upload(p, v, m)
        f = e.compute_forces()
        cnt = e.interaction_counts()
    d = O.compute_forces_diag(O.build_tree(p, m, md if md < 32 else 0), p, m, compat_self_skip=compat)
    ok = np.isfinite(d.forces).all(axis=1)                                     # (exactly coincident bodies: NaN in the reference)
    assert ok.mean() > 0.999
    assert rel(f[ok], d.forces[ok]).max() <= TOL
    assert np.array_equal(cnt[ok], d.counts[ok])


@pytest.mark.parametrize("kind,n,theta", [("plummer", 65536, 0.5), ("uniform", 65536, 0.3), ("uniform", 257, 0.5),
                                          ("uniform", 2, 0.5), ("uniform", 1, 0.5)])
def test_synthetic_distributions_against_the_uncapped_oracle(kind, n, theta):
    m, p, v = IC.make(kind, n, 5, quasi_static=True)
    with engine(n, max_depth=32, theta=theta, reference_compat=False, flags=FLAG_WALK_STATS) as e:
        e.upload(p, v, m)
        f = e.compute_forces()
        cnt = e.interaction_counts()
    d = O.compute_forces_diag(O.build_tree(p, m, 0), p, m, theta=theta, compat_self_skip=False)
    if n == 1:
        assert np.array_equal(f, np.zeros((1, 2))) and cnt[0] == 0
        return
    assert rel(f, d.forces).max() <= TOL
    assert np.array_equal(cnt, d.counts)


def test_agrees_with_the_bit_exact_mode_and_is_deterministic(gold):
    """Same tree, same node values: the two fp64 walks differ by the order of a body's sum and the rsqrt only."""
    g = gold("ref_project_40960")
    m, p, v = g["mass"], g["pos"], g["vel"]
    out = []
    for prec in (G.Precision.F64_EXACT, G.Precision.F64, G.Precision.F64):
        with engine(40960, precision=prec) as e:
            e.upload(p, v, m)
            out.append(e.compute_forces())
    assert np.array_equal(out[0], g["forces_0"])                                # (the anchor is the golden one)
    assert rel(out[1], out[0]).max() <= TOL
    assert np.array_equal(out[1], out[2])


def test_short_trajectory_on_the_encounter_free_fixture(gold):
    g = gold("ref_project_4096_grid")
    m, p, v = g["mass"], g["pos"], g["vel"]
    po, vo = O.run(p, v, m, 10, max_depth=10)
    with engine(4096) as e:
        e.upload(p, v, m)
        e.step(10)
        pg, vg = e.download()
    box = np.ptp(p, axis=0).max()
    assert np.abs(pg - po).max() <= 1e-11 * box
    assert np.abs(vg - vo).max() <= 1e-11 * np.abs(vo).max()


def test_full_size_slice_and_properties():
    """BASELINE config 3's size: N = 1,048,576 Plummer, theta 0.5, uncapped (max_depth 21 like the fp32 headline)."""
    n = 1 << 20
    m, p, v = IC.make("plummer", n, 1, quasi_static=True)
    with engine(n, max_depth=21, reference_compat=True, flags=FLAG_WALK_STATS) as e:
        e.upload(p, v, m)
        f = e.compute_forces()
        cnt = e.interaction_counts()
        st = e.stats()
        e.step(2)
        pp, _ = e.download()
    assert np.isfinite(pp).all()
    t = O.build_tree(p, m, 21)
    assert st.n_nodes == len(t)
    lo, hi = 400000, 465536
    d = O.compute_forces_diag(t, p, m, compat_self_skip=True, lo=lo, hi=hi)
    ok = np.isfinite(d.forces[lo:hi]).all(axis=1)
    assert rel(f[lo:hi][ok], d.forces[lo:hi][ok]).max() <= TOL
    assert np.array_equal(cnt[lo:hi][ok], d.counts[lo:hi][ok])


@pytest.mark.parametrize("kind,n,md,compat,theta", [("plummer", 20000, 21, False, 0.5), ("uniform", 65536, 21, True, 0.5),
                                                    ("clumped", 30000, 8, False, 0.5), ("clumped", 30000, 8, True, 0.5),
                                                    ("plummer", 30000, 32, False, 0.3), ("clumped", 20000, 32, True, 0.5),
                                                    ("uniform", 130, 4, True, 0.5), ("uniform", 1, 21, True, 0.5)])
def test_asm_walk_equals_the_portable_walk(kind, n, md, compat, theta):
    """The hand-written gfx950 traversal loop (walk64_asm, csrc/bh_walk_f64.hpp) performs the same operations in the same
    order as the C++ loop beside it (BH_FLAG_WALK_PORTABLE; BH_FLAG_WALK_STATS runs that loop with counters): forces after
    one evaluation and the state after 3 steps are BITWISE equal -- subdivided cells, leaves and the occupant test,
    depth-cap aggregates with the reference's occupant code (compat on), the empty-node cut-off, the two-tier stack of
    trees deeper than 21 levels (max_depth 32), a ragged last wave, a one-body tree (the loop is never entered)."""
    if kind == "clumped":
        rng = np.random.default_rng(5)
        p = np.concatenate([rng.normal(0, 1e-3, (n // 2, 2)), rng.uniform(-1, 1, (n - n // 2, 2))])
        m, v = rng.uniform(0.1, 0.5, n), rng.uniform(-1e-9, 1e-9, (n, 2))
        m[::7] = 1e-16                                                          # below the reference's 1e-15 cut-off
    else:
        m, p, v = IC.make(kind, n, 3, quasi_static=True)
    res = []
    for flags in (0, FLAG_WALK_PORTABLE, FLAG_WALK_STATS):
        with engine(n, max_depth=md, reference_compat=compat, theta=theta, flags=flags) as e:
            e.upload(p, v, m)
            f = e.compute_forces()
            e.step(3)
            res.append((f,) + e.download())
    if n > 1:
        assert np.nanmax(np.abs(res[0][0])) > 0
    for other in res[1:]:
        for x, y in zip(res[0], other):
            assert np.array_equal(x, y, equal_nan=True)


def test_random_small_systems_against_the_oracle():
    """60 seeded random systems (2 to 600 bodies; clusters, masses 1e-16 to 100, depth caps 2 to 32, theta 0.05 to 2, both
    occupant rules): the throughput walk's forces within 1e-12 of the oracle's + the forward rounding model of the
    coordinate differences (1e-13 x WalkDiag.coord, the force goes with d^-3: a body that shares a depth-cap cell with a 1e-16-mass neighbour meets an
    aggregate 1e-21 away from itself, and its "force" of 1e20 is rounding noise in the reference as well), with the oracle's
    per-body interaction counts -- both the hand-written loop and its counting statement."""
    rng = np.random.default_rng(91)
    for case in range(60):
        n = int(rng.integers(2, 600))
        md = int(rng.choice([2, 4, 7, 10, 15, 21, 32]))
        theta = float(10.0 ** rng.uniform(-1.3, 0.3))
        compat = bool(rng.integers(0, 2))
        centres = rng.uniform(-1, 1, (int(rng.integers(1, 5)), 2))
        p = centres[rng.integers(0, len(centres), n)] + rng.normal(0, 10.0 ** rng.uniform(-4, -1), (n, 2))
        m = 10.0 ** rng.uniform(-4, 2, n)
        m[rng.random(n) < 0.08] = 1e-16
        v = np.zeros((n, 2))
        # (pos_rounded: the model counts leaves too -- a depth-cap aggregate is a leaf of the reference's tree)
        d = O.compute_forces_diag(O.build_tree(p, m, md), p, m, theta=theta, compat_self_skip=compat, pos_rounded=True)
        fo = d.forces
        # (a term whose distance is 1e-8 of the coordinates it is the difference of -- a body against the depth-cap aggregate
        #  it is itself most of -- has no significant digits in the reference either: those bodies are not compared)
        ok = np.isfinite(fo).all(axis=1) & np.isfinite(d.coord) & (d.coord < 1e8 * d.abs_sum)
        for flags in (0, FLAG_WALK_STATS):
            with engine(n, theta=theta, max_depth=md, reference_compat=compat, flags=flags, node_capacity=140 * n + 4096) as e:
                e.upload(p, v, m)
                f = e.compute_forces()
                if flags:
                    assert np.array_equal(e.interaction_counts()[ok], d.counts[ok]), (case, n, md, theta, compat)
            err = np.linalg.norm(f[ok] - fo[ok], axis=1)
            tol = TOL * np.linalg.norm(fo[ok], axis=1) + 1e-13 * d.coord[ok]
            assert (err <= tol).all(), (case, n, md, theta, compat, float((err / np.maximum(tol, 1e-300)).max()))
