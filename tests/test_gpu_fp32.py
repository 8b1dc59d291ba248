"""GPU parity, fp32 mode (BASELINE configs "fp32"): libbhgpu vs the fp64 oracle on the SAME inputs
(float32-representable values, so both sides see identical bodies).

Stated tolerances (north_star: "within a stated fp32 tolerance"), per body, relative L2 error of the
acceleration |a_gpu - a_oracle| / |a_oracle|:
    median <= 2e-6 ; 99.9 % of bodies <= 1e-4 ; every body <= 5e-3
The tail is NOT rounding noise: the MAC `size/dist < theta` is evaluated in fp32, so for a node
within ~1e-7 of the threshold a body may open it where the oracle accepts it (or vice versa),
which changes that one term by its multipole error.  Positions after k steps: <= 1e-6 x box width.
The tree itself (topology, occupants) must be IDENTICAL to the oracle's; COM/mass to fp32 rounding."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import bh_oracle as O  # noqa: E402
import gpu_nbody_simulation_amd as G  # noqa: E402
from gpu_nbody_simulation_amd import initial_conditions as IC  # noqa: E402
from gpu_nbody_simulation_amd.engine import (FLAG_LDS_STACK, FLAG_WALK_NO_SPLIT, FLAG_WALK_PORTABLE,  # noqa: E402
                                             FLAG_WALK_STATS)

TOL_MEDIAN, TOL_P999, TOL_MAX = 2e-6, 1e-4, 5e-3


def f32(a):
    return np.asarray(a, dtype=np.float64).astype(np.float32).astype(np.float64)


def rel_err(a, ref):
    return np.linalg.norm(a - ref, axis=1) / np.linalg.norm(ref, axis=1)


def check_tolerance(a, ref):
    r = rel_err(a, ref)
    assert np.median(r) <= TOL_MEDIAN, np.median(r)
    assert np.quantile(r, 0.999) <= TOL_P999, np.quantile(r, 0.999)
    assert r.max() <= TOL_MAX, r.max()


def engine(n, **kw):
    kw.setdefault("precision", G.Precision.F32)
    return G.BarnesHutEngine(G.BhConfig(capacity=n, **kw))


def test_tree_topology_is_the_oracles(gold):
    g = gold("ref_project_4096_grid")                        # inputs are float32-representable
    m, p, v = g["mass"], g["pos"], g["vel"]
    with engine(4096, max_depth=10) as e:
        e.upload(p, v, m)
        e.build_tree()
        nodes, depth = e.export_tree()
    rn, rd = O.canonical_tree(O.build_tree(p, m, 10))
    assert len(nodes) == len(rn) and np.array_equal(depth, rd)
    for f in ("xmin", "xmax", "ymin", "ymax", "particle"):   # fp64 bisection -> bitwise
        assert np.array_equal(nodes[f], rn[f]), f
    assert np.array_equal(nodes["child"] == -1, rn["child"] == -1)
    assert np.allclose(nodes["mass"], rn["mass"], rtol=3e-7, atol=0)
    assert np.allclose(nodes["comx"], rn["comx"], rtol=0, atol=3e-8)    # 0.1 * 2^-23 * few
    assert np.allclose(nodes["comy"], rn["comy"], rtol=0, atol=3e-8)


@pytest.mark.parametrize("flags", [0, FLAG_LDS_STACK])
def test_accelerations_encounter_free_case(gold, flags):
    g = gold("ref_project_4096_grid")
    m, p, v = g["mass"], g["pos"], g["vel"]
    ref = g["forces_0"] / m[:, None]                         # the REFERENCE's forces (golden)
    with engine(4096, max_depth=10, flags=flags | FLAG_WALK_STATS) as e:
        e.upload(p, v, m)
        e.compute_forces()
        a = e.accelerations()
        st = e.stats()
    check_tolerance(a, ref)
    _, ws = O.compute_forces(O.build_tree(p, m, 10), p, m, with_stats=True)
    assert abs(st.interactions - ws.interactions) <= 1e-4 * ws.interactions   # MAC flips only


def test_lds_and_register_stacks_agree_bitwise(gold):
    g = gold("ref_project_40960")
    m, p, v = f32(g["mass"]), f32(g["pos"]), f32(g["vel"])
    out = []
    # (one wavefront per 64 bodies on both sides: the split walk exists for the register stack only)
    for flags in (FLAG_WALK_NO_SPLIT, FLAG_LDS_STACK):
        with engine(40960, max_depth=16, flags=flags) as e:
            e.upload(p, v, m)
            e.step(3)
            out.append(e.download())
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


@pytest.mark.parametrize("kind,n,md,compat", [("plummer", 40000, 21, False), ("uniform", 20001, 16, False),
                                              ("clumped", 30000, 8, False), ("clumped", 30000, 8, True),
                                              ("plummer", 300000, 21, False)])
def test_asm_walk_equals_the_portable_walk(kind, n, md, compat):
    """The hand-scheduled gfx950 traversal loop (the default for one wavefront per 64 bodies) performs the
    same operations in the same order as the C++ loop (BH_FLAG_WALK_PORTABLE): accelerations after one
    force evaluation and the state after 3 steps are BITWISE equal -- subdivided cells, leaves, the self
    skip, bucket leaves (shallow cap, compat off), depth-cap aggregates (compat on), ragged last wave."""
    if kind == "clumped":
        rng = np.random.default_rng(5)
        p = f32(np.concatenate([rng.normal(0, 1e-3, (n // 2, 2)), rng.uniform(-1, 1, (n - n // 2, 2))]))
        m, v = f32(rng.uniform(0.1, 0.5, n)), f32(rng.uniform(-1e-9, 1e-9, (n, 2)))
    else:
        m, p, v = IC.make(kind, n, 3, quasi_static=True)
    res = []
    for flags in (FLAG_WALK_NO_SPLIT, FLAG_WALK_NO_SPLIT | FLAG_WALK_PORTABLE):
        with engine(n, max_depth=md, reference_compat=compat, flags=flags) as e:
            e.upload(p, v, m)
            e.compute_forces()
            a = e.accelerations()
            e.step(3)
            res.append((a,) + e.download())
    assert np.isfinite(res[0][0]).all() and np.abs(res[0][0]).max() > 0
    for x, y in zip(res[0], res[1]):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("kind,n,md,compat", [("plummer", 20000, 21, False), ("uniform", 65536, 21, False),
                                              ("clumped", 30000, 8, False), ("clumped", 30000, 8, True),
                                              ("plummer", 100000, 21, False), ("uniform", 130, 4, False)])
def test_asm_list_walk_equals_the_portable_level_walk(kind, n, md, compat):
    """Small launches walk level by level with several wavefronts per group (8 up to 768 groups, 4 up to
    3,072); the per-chunk evaluation has the same hand-scheduled child blocks (walk_list_asm).  Bitwise
    equal to the C++ loop (BH_FLAG_WALK_PORTABLE) -- accelerations and a 3-step trajectory."""
    if kind == "clumped":
        rng = np.random.default_rng(5)
        p = f32(np.concatenate([rng.normal(0, 1e-3, (n // 2, 2)), rng.uniform(-1, 1, (n - n // 2, 2))]))
        m, v = f32(rng.uniform(0.1, 0.5, n)), f32(rng.uniform(-1e-9, 1e-9, (n, 2)))
    else:
        m, p, v = IC.make(kind, n, 3, quasi_static=True)
    res = []
    for flags in (0, FLAG_WALK_PORTABLE):
        with engine(n, max_depth=md, reference_compat=compat, flags=flags) as e:
            e.upload(p, v, m)
            e.compute_forces()
            a = e.accelerations()
            e.step(3)
            res.append((a,) + e.download())
    assert np.isfinite(res[0][0]).all() and np.abs(res[0][0]).max() > 0
    for x, y in zip(res[0], res[1]):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("kind,n,md", [("plummer", 200000, 21), ("clumped", 30000, 8), ("uniform", 513, 3)])
def test_packed_sort_equals_the_two_array_sort(monkeypatch, kind, n, md):
    """Keys of <= 40 bits carry the body index in the key word through the radix passes (one 8-byte array per
    pass instead of key + index arrays; BH_SORT_PACK=0 restores the two arrays).  Same stable order, so the
    tree, the forces and a moving trajectory are BITWISE the same -- equal keys (bodies sharing a depth-cap
    cell: the clumped case) keep their body order either way."""
    if kind == "clumped":
        rng = np.random.default_rng(5)
        p = f32(np.concatenate([rng.normal(0, 1e-3, (n // 2, 2)), rng.uniform(-1, 1, (n - n // 2, 2))]))
        m, v = f32(rng.uniform(0.1, 0.5, n)), f32(rng.normal(0, 1e-4, (n, 2)))
    else:
        m, p, v = IC.make(kind, n, 3, quasi_static=True, drift_cells=1.0)
    res = []
    for packv in ("1", "0"):
        monkeypatch.setenv("BH_SORT_PACK", packv)
        with engine(n, max_depth=md, reference_compat=False) as e:
            e.upload(p, v, m)
            e.build_tree()
            nodes, depth = e.export_tree()
            e.step(20)                                             # crosses a re-ordering of the state (every 16th build)
            res.append((nodes, depth) + e.download())
    for x, y in zip(res[0], res[1]):
        assert np.array_equal(x, y)


def _sort_case(kind, n):
    if kind == "clumped":
        rng = np.random.default_rng(5)
        p = f32(np.concatenate([rng.normal(0, 1e-3, (n // 2, 2)), rng.uniform(-1, 1, (n - n // 2, 2))]))
        return f32(rng.uniform(0.1, 0.5, n)), p, f32(rng.normal(0, 1e-4, (n, 2)))
    return IC.make(kind, n, 3, quasi_static=True, drift_cells=1.0)


@pytest.mark.parametrize("kind,n,md,precision", [("plummer", 200000, 21, "f32"), ("clumped", 30000, 8, "f32"),
                                                  ("uniform", 513, 3, "f32"), ("uniform", 2, 5, "f32"),
                                                  ("plummer", 1100000, 21, "f32"), ("uniform", 70000, 21, "mixed"),
                                                  ("clumped", 60000, 21, "f32"), ("plummer", 2500000, 21, "f32"),
                                                  ("uniform", 150000, 21, "exact"), ("uniform", 40000, 10, "exact"),
                                                  ("uniform", 300000, 21, "f64")])
def test_bucket_sort_equals_the_lsd_sort(monkeypatch, kind, n, md, precision):
    """From the second build on the keys are sorted by ONE counting pass over 256 buckets -- splitters = the
    previous build's sorted positions at every n/256-th rank, re-keyed in the new root box -- and an in-LDS
    sort of every bucket (BH_SORT_BUCKET=0: the five LSD passes every time).  Same stable order: the tree of
    a later build and a moving trajectory are BITWISE the same, including bodies that share a depth-cap
    cell (equal keys keep body order) and buckets of equal keys larger than the LDS capacity.  (2.5M bodies:
    1,024 buckets, 10-bit counting pass.)"""
    m, p, v = _sort_case(kind, n)
    # (the exact modes since round 3: their state stays in caller order, their samples come through the previous perm)
    prec = {"f32": G.Precision.F32, "mixed": G.Precision.MIXED, "exact": G.Precision.F64_EXACT, "f64": G.Precision.F64}[precision]
    res, spills = [], []
    for mode in ("1", "0"):
        monkeypatch.setenv("BH_SORT_BUCKET", mode)
        # (exact modes: the reference's own self skip, or a lone body in a depth-cap cell meets itself at distance 0)
        with engine(n, max_depth=md, reference_compat=precision in ("exact", "f64"), precision=prec) as e:
            e.upload(p, v, m)
            e.step(3)
            e.build_tree()                                         # (mode 1: a bucket-sorted build)
            nodes, depth = e.export_tree()
            # (f32 / mixed: crosses a re-ordering of the state, every 16th build; the exact modes never re-order, and with
            #  the reference's arithmetic a drifting cloud meets its first inf * 0 within twenty steps: NaN == NaN proves nothing)
            e.step(18 if precision in ("f32", "mixed") else 4)
            res.append((nodes, depth) + e.download())
            spills.append(e.stats().sort_spill_buckets)
    for x, y in zip(res[0], res[1]):
        assert np.array_equal(x, y)
    assert np.isfinite(res[0][2]).all()                            # (a trajectory that blew up would compare NaN with NaN)
    assert spills[1] == 0
    if kind != "clumped":
        assert spills[0] == 0                                      # steady motion: every bucket fits on chip


@pytest.mark.parametrize("precision", ["f32", "exact", "f64"])
@pytest.mark.parametrize("n,md", [(2, 5), (63, 10), (1024, 10), (1025, 21), (4096, 4), (4096, 21), (4097, 21)])
def test_small_launches_are_one_bucket(monkeypatch, n, md, precision):
    """Round 4: a launch of at most 4,096 bodies is sorted by bucket_sort_kernel alone -- one workgroup, the keys as
    keys_kernel left them, no splitters and no counting pass, from the first build on.  Same stable order as the LSD passes
    (BH_SORT_BUCKET=0) and as the splitter path (=2): tree, forces and a moving trajectory BITWISE the same, depth-cap cells
    that hold many equal keys included (max_depth 4 at 4,096 bodies: 64 cells of 64)."""
    m, p, v = _sort_case("uniform", n)
    prec = {"f32": G.Precision.F32, "exact": G.Precision.F64_EXACT, "f64": G.Precision.F64}[precision]
    res = []
    for mode in ("1", "0", "2"):
        monkeypatch.setenv("BH_SORT_BUCKET", mode)
        with engine(n, max_depth=md, reference_compat=precision != "f32", precision=prec) as e:
            e.upload(p, v, m)
            f = e.compute_forces()                                 # (mode 1: one bucket already in the first build)
            nodes, depth = e.export_tree()
            e.step(5)
            e.build_tree()
            nodes2, depth2 = e.export_tree()
            res.append((f, nodes, depth, nodes2, depth2) + e.download())
            assert e.stats().sort_spill_buckets == 0
    for other in res[1:]:
        for x, y in zip(res[0], other):
            if x.dtype.names:                                      # (the exported tree: a record array)
                assert all(np.array_equal(x[f], y[f], equal_nan=True) for f in x.dtype.names)
            else:
                assert np.array_equal(x, y, equal_nan=True)
    assert np.isfinite(res[0][5]).all()


def test_bucket_sort_with_small_build_tiles_above_1m_bodies(monkeypatch):
    """BH_BUILD_ITEMS=2 (512-key tiles) is honoured up to 4M bodies, and between 1M and 4M the bucket
    pass counts 1,024 buckets per tile -- 2 words per body, which the counting scratch (sized for 2,048-key tiles)
    did not hold.  Same stable order as the LSD passes with the same tiles, no spill."""
    n = 2_000_000
    m, p, v = _sort_case("plummer", n)
    monkeypatch.setenv("BH_BUILD_ITEMS", "2")
    res = []
    for mode in ("1", "0"):
        monkeypatch.setenv("BH_SORT_BUCKET", mode)
        with engine(n, max_depth=21, reference_compat=False) as e:
            e.upload(p, v, m)
            e.step(4)
            e.build_tree()
            nodes, depth = e.export_tree()
            res.append((nodes, depth) + e.download())
            assert e.stats().sort_spill_buckets == 0
    for x, y in zip(res[0], res[1]):
        assert np.array_equal(x, y)


def test_bucket_sort_with_useless_splitters_spills_and_stays_correct(monkeypatch):
    """BH_SORT_BUCKET=2 takes the splitters from whatever the last build left behind even after bh_upload
    replaced the bodies: a uniform box first, then a tight clump with two far bodies -- nearly all keys fall
    into a few buckets, which their workgroups sort through memory.  The order is still the LSD sort's."""
    n = 50000
    m1, p1, v1 = IC.make("uniform", n, 9, quasi_static=True)
    rng = np.random.default_rng(11)
    p2 = f32(np.concatenate([rng.normal(0.3, 2e-3, (n - 2, 2)), [[-1.0, -1.0], [1.0, 1.0]]]))
    v2 = f32(rng.normal(0, 1e-5, (n, 2)))
    res = []
    for mode in ("2", "0"):
        monkeypatch.setenv("BH_SORT_BUCKET", mode)
        with engine(n, max_depth=21, reference_compat=False) as e:
            e.upload(p1, v1, m1)
            e.step(2)
            before = e.stats().sort_spill_buckets
            e.upload(p2, v2, m1)
            e.build_tree()
            nodes, depth = e.export_tree()
            spilled = e.stats().sort_spill_buckets - before
            e.step(5)
            res.append((nodes, depth) + e.download())
            if mode == "2":
                assert spilled >= 1
                assert e.stats().sort_spill_buckets - before == spilled   # the next builds are balanced again
    for x, y in zip(res[0], res[1]):
        assert np.array_equal(x, y)


def test_bucket_sort_reruns_a_bucket_whose_keys_crowd_below_a_wide_span(monkeypatch):
    """A bucket whose keys span more than 24 bits is sorted by the top 24 bits of that span in three byte passes and
    finished by counting inside every run of equal top bits (csrc/bh_sort.hpp) -- runs are single keys unless bodies
    crowd: here 48 bodies sit within a few depth-20 cells of a uniform box, a run of 48 keys that differ in their
    lowest bits only.  Its workgroup notices (runs are followed for 8 keys), sorts the bucket again with all byte
    passes and counts that (bh_stats.sort_rerun_buckets); the order is the LSD sort's bit for bit either way, and a
    box without the crowd reruns nothing."""
    n = 20000
    rng = np.random.default_rng(21)
    base = f32(rng.uniform(-1, 1, (n, 2)))
    crowd = base.copy()
    crowd[:48] = f32(np.array([0.3712, -0.2288]) + rng.uniform(0, 1.4e-5, (48, 2)))
    m, v = f32(rng.uniform(0.5, 1.5, n)), f32(rng.normal(0, 1e-7, (n, 2)))
    for p, want_rerun in ((crowd, True), (base, False)):
        res, reruns = [], []
        for mode in ("1", "0"):
            monkeypatch.setenv("BH_SORT_BUCKET", mode)
            with engine(n, max_depth=21, reference_compat=False) as e:
                e.upload(p, v, m)
                e.step(3)
                e.build_tree()
                nodes, depth = e.export_tree()
                e.step(3)
                res.append((nodes, depth) + e.download())
                st = e.stats()
                reruns.append(st.sort_rerun_buckets)
                assert st.sort_spill_buckets == 0
        for x, y in zip(res[0], res[1]):
            assert np.array_equal(x, y)
        assert reruns[1] == 0 and (reruns[0] >= 1) == want_rerun, reruns


def test_bucket_sort_with_bodies_piled_into_one_cell(monkeypatch):
    """Bodies that collapse into a few depth-cap cells (what close encounters without softening do to the
    reference's runs: the root box blows up) share their keys.  A key value frequent enough to be sampled
    twice gets a bucket of its own -- equal keys need no sorting, whatever their number -- so the build stays
    the LSD sort's bit for bit and nothing is sorted through memory, step after step."""
    n = 60000
    rng = np.random.default_rng(3)
    p = f32(np.concatenate([rng.normal(0.25, 1e-7, (40000, 2)), rng.normal(-0.5, 1e-7, (15000, 2)),
                            rng.uniform(-1, 1, (n - 55000, 2))]))
    m, v = f32(rng.uniform(1e-14, 2e-14, n)), f32(rng.normal(0, 1e-9, (n, 2)))
    res, spills = [], []
    for mode in ("1", "0"):
        monkeypatch.setenv("BH_SORT_BUCKET", mode)
        with engine(n, max_depth=12, reference_compat=True) as e:
            e.upload(p, v, m)
            e.step(6)
            e.build_tree()
            nodes, depth = e.export_tree()
            e.step(6)
            res.append((nodes, depth) + e.download())
            spills.append(e.stats().sort_spill_buckets)
    for x, y in zip(res[0], res[1]):
        assert np.array_equal(x, y)
    assert spills == [0, 0]


def test_bucket_sort_survives_an_exploding_root_box(monkeypatch):
    """The reference's own regime: heavy bodies, no softening, dt = 1 -- close encounters eject bodies, the root
    box grows by orders of magnitude within two steps and the cloud shrinks into a corner of it (the tree goes
    from 58,000 to 485 nodes).  The splitters are re-keyed in every build's box at full depth, so the buckets
    stay balanced through the explosion: same trajectory as with the LSD passes, bit for bit, and no bucket is
    sorted through memory."""
    n = 20000
    r = np.random.default_rng(0)
    m, p, v = f32(10.0 ** r.uniform(-2, 1, n)), f32(r.uniform(-0.1, 0.1, (n, 2))), f32(r.uniform(-1e-4, 1e-4, (n, 2)))
    res, spills, nodes = [], [], []
    for mode in ("1", "0"):
        monkeypatch.setenv("BH_SORT_BUCKET", mode)
        with engine(n, max_depth=16) as e:
            e.upload(p, v, m)
            e.step(2)
            nodes.append(e.stats().n_nodes)
            e.step(6)
            nodes.append(e.stats().n_nodes)
            res.append(e.download())
            spills.append(e.stats().sort_spill_buckets)
    assert nodes[0] > 20000 and nodes[1] < 2000                      # the explosion happened
    assert np.array_equal(res[0][0], res[1][0], equal_nan=True) and np.array_equal(res[0][1], res[1][1], equal_nan=True)
    assert spills == [0, 0]


def test_multistep_trajectory_encounter_free_case(gold):
    """20 steps vs the REFERENCE's own trajectory (golden; encounter-free by construction, see
    scripts/make_golden.py): positions <= 1e-6 x box width; the velocity CHANGE (what the forces did)
    within 1e-3 relative for 99 % of the bodies."""
    g = gold("ref_project_4096_grid")
    m, p, v = g["mass"], g["pos"], g["vel"]
    with engine(4096, max_depth=10) as e:
        e.upload(p, v, m)
        e.step(20)
        pp, vv = e.download()
    box = 0.24
    assert np.abs(pp - g["pos_after_19"]).max() <= 1e-6 * box
    dv_ref = g["vel_after_19"] - v
    dv = vv - v
    r = np.linalg.norm(dv - dv_ref, axis=1) / np.linalg.norm(dv_ref, axis=1)
    assert np.quantile(r, 0.99) <= 1e-3 and np.median(r) <= 1e-4


def test_reference_compat_aggregates_cap_cells(gold):
    """compat on, cap 10, shipped files: same semantics as project.cu (aggregate + artefact).  The
    bodies that sit in multi-occupant cap cells interact with their own cell's COM at tiny distance,
    which amplifies fp32 rounding; they are compared separately."""
    g = gold("ref_project_40960")
    m, p, v = f32(g["mass"]), f32(g["pos"]), f32(g["vel"])
    t = O.build_tree(p, m, 10)
    ref = O.compute_forces(t, p, m) / m[:, None]
    with engine(40960, max_depth=10, reference_compat=True) as e:
        e.upload(p, v, m)
        e.compute_forces()
        a = e.accelerations()
        nodes, depth = e.export_tree()
    assert len(nodes) == len(t)
    r = rel_err(a, ref)
    assert np.median(r) <= TOL_MEDIAN
    assert np.quantile(r, 0.75) <= 1e-5


@pytest.mark.parametrize("kind,n,md", [("uniform", 65536, 21), ("plummer", 65536, 21), ("plummer", 32768, 14)])
def test_bucket_mode_matches_uncapped_oracle(kind, n, md):
    """compat off: a depth-cap cell holding several bodies is summed body by body, which is what the
    uncapped tree of main_approach_2.cpp converges to.  md=14 forces many buckets."""
    m, p, v = IC.make(kind, n, 2)
    t = O.build_tree(p, m, 0)
    ref = O.compute_forces(t, p, m, compat_self_skip=False) / m[:, None]
    with engine(n, max_depth=md, reference_compat=False, flags=FLAG_WALK_STATS) as e:
        e.upload(p, v, m)
        e.compute_forces()
        a = e.accelerations()
    assert np.isfinite(a).all()
    r = rel_err(a, ref)
    if md >= 21:
        check_tolerance(a, ref)
    else:   # buckets replace subtrees by direct sums: MORE accurate than the oracle's multipoles,
            # so compare with the direct sum on a sample instead
        idx = np.arange(0, n, 64)
        d = O.direct_forces(p, m)[idx] / m[idx, None] if n <= 32768 else None
        assert np.median(rel_err(a[idx], d)) < 2e-2
        assert np.median(r) < 1e-2


@pytest.mark.parametrize("flags", [0, FLAG_LDS_STACK, FLAG_WALK_PORTABLE])
def test_deep_trees_fit_the_stack(flags):
    """max_depth 32: the 128-entry register-lane stack pairs entries only while sp <= 120 - 3 * 31 = 27, which
    keeps every wavefront inside its bound (walk_tree_asm); the asm loop, the C++ loop and the LDS-stack
    variant agree bitwise.  A clumped input makes the tree really deep."""
    rng = np.random.default_rng(8)
    n = 16384
    p = f32(np.concatenate([rng.normal(0, 1e-5, (n // 4, 2)), rng.normal(0, 1e-3, (n // 4, 2)), rng.uniform(-1, 1, (n // 2, 2))]))
    m, v = f32(rng.uniform(0.1, 0.5, n)), np.zeros((n, 2))
    t = O.build_tree(p, m, 0)
    ref = O.compute_forces(t, p, m, compat_self_skip=False) / m[:, None]
    with engine(n, max_depth=32, reference_compat=False, flags=flags | FLAG_WALK_NO_SPLIT) as e:
        e.upload(p, v, m)
        e.compute_forces()
        a = e.accelerations()
        nodes, depth = e.export_tree()
    assert depth.max() >= 23
    check_tolerance(a, ref)
    if flags:
        with engine(n, max_depth=32, reference_compat=False, flags=FLAG_WALK_NO_SPLIT) as e:
            e.upload(p, v, m)
            e.compute_forces()
            assert np.array_equal(a, e.accelerations())


def test_coincident_bodies_do_not_poison_the_run():
    """fp32 positions are quantised: two bodies can land on the same float.  The reference divides by
    zero there (NaN, project.cu:651-658); fp32 mode makes a zero-distance pair exert no force."""
    m, p, v = IC.make("uniform", 4096, 4)
    p[100] = p[101]
    p[200:204] = p[200]
    for compat in (True, False):
        with engine(4096, max_depth=21, reference_compat=compat) as e:
            e.upload(p, v, m)
            e.step(5)
            pp, vv = e.download()
            assert np.isfinite(pp).all() and np.isfinite(vv).all()


def test_full_size_properties():
    """BASELINE metric size (N = 1,048,576, Plummer, theta 0.5): size-independent properties."""
    n = 1 << 20
    m, p, v = IC.make("plummer", n, 1, quasi_static=True)
    with engine(n, max_depth=21, reference_compat=False, flags=FLAG_WALK_STATS) as e:
        e.upload(p, v, m)
        e.compute_forces()
        a1 = e.accelerations()
        st = e.stats()
        e.compute_forces()
        a2 = e.accelerations()
        # oracle on a slice of the same bodies (the full CPU walk takes ~8 s; 4,096 bodies here)
        t = O.build_tree(p, m, 0)
        ref = O.compute_forces(t, p, m, compat_self_skip=False, lo=500000, hi=504096)[500000:504096] / m[500000:504096, None]
        check_tolerance(a1[500000:504096], ref)
    assert np.array_equal(a1, a2)                                   # deterministic
    assert np.isfinite(a1).all()
    assert st.n_nodes == 1 + 4 * st.n_internal
    assert st.n_nodes == len(O.build_tree(p, m, 21))               # same tree as the depth-21 oracle
    # Newton's third law: the mass-weighted accelerations cancel up to the multipole error
    net = np.abs((m[:, None] * a1).sum(0)).max()
    scale = (m[:, None] * np.abs(a1)).sum()
    assert net <= 2e-3 * scale
    assert 300 < st.interactions / n < 600 and 10 < st.wave_nodes / n < 25


def test_direct_sum_limit_against_main_approach_1(gold, init1024):
    """BASELINE config[0] names main_approach_1.cpp (O(N^2) direct sum, fp64).  With theta -> 0 every
    cell is opened and the fp32 walk degenerates to the direct sum: forces of step 0 within fp32
    rounding of the REFERENCE's; the 10-step trajectory within the stated percentiles."""
    m, p, v = init1024
    g = gold("ref_ma1_1024")
    with engine(1024, max_depth=21, theta=1e-6, reference_compat=False) as e:
        e.upload(p, v, m)
        e.compute_forces()
        f = e.forces()
        e.step(10)
        pp, vv = e.download()
    r = rel_err(f, g["forces_0"])
    assert np.median(r) < 5e-6 and r.max() < 5e-4
    # 10 steps.  The shipped bodies include pairs 1e-5 apart; fp32 positions resolve
    # 7e-9, so the relative error of such a pair's separation -- and of its mutual force -- is ~1e-3 and
    # grows with every encounter: the tail is physics + precision, not a bug.  Measured on MI355X:
    # position error quantiles 50/90/99/max = 1.5e-8 / 1.6e-6 / 5.1e-5 / 7.9e-4.
    err = np.abs(pp - g["pos_after_9"]).max(axis=1)
    assert np.median(err) <= 1e-7 and np.quantile(err, 0.9) <= 1e-5
    assert np.quantile(err, 0.99) <= 3e-4 and err.max() <= 5e-3
    dv, dv_ref = vv - v, g["vel_after_9"] - v
    rv = np.linalg.norm(dv - dv_ref, axis=1) / np.linalg.norm(dv_ref, axis=1)
    assert np.median(rv) < 1e-4 and np.quantile(rv, 0.9) < 5e-3


def test_degenerate_input_stays_bounded():
    """One body at infinity collapses every key into one depth-cap cell.  With compat off that cell
    would be a bucket of N bodies (an O(N^2) step that looks like a hang); cells above 1,024 bodies
    are aggregated instead, so the step returns promptly and finitely for the regular bodies."""
    import time
    n = 200000
    m, p, v = IC.make("uniform", n, 6)
    p[17] = [np.inf, 0.0]
    with engine(n, max_depth=21, reference_compat=False) as e:
        e.upload(p, v, m)
        t0 = time.time()
        e.step(2)
        e.sync()
        assert time.time() - t0 < 5.0
        assert e.stats().n_nodes <= 1 + 4 * 21


@pytest.mark.parametrize("split", [2, 4, 8, 0])
@pytest.mark.parametrize("kind,n,md", [("plummer", 65536, 21), ("clumped", 30000, 8)])
def test_split_walk_equals_the_one_wave_walk(monkeypatch, split, kind, n, md):
    """Several wavefronts per 64-body group (launches of few bodies; BH_WALK_SPLIT forces the factor,
    0 = the automatic choice): same node set, same per-body MAC, only the order of the fp32 sums
    changes -> identical interaction counts, accelerations equal to summation rounding, and the
    per-group min/max left for the next step's root box are exact."""
    if kind == "clumped":                                     # shallow cap: multi-body bucket leaves
        rng = np.random.default_rng(5)
        p = f32(np.concatenate([rng.normal(0, 1e-3, (n // 2, 2)), rng.uniform(-1, 1, (n - n // 2, 2))]))
        m, v = f32(rng.uniform(0.1, 0.5, n)), f32(rng.uniform(-1e-9, 1e-9, (n, 2)))
    else:
        m, p, v = IC.make(kind, n, 3)
        m, p, v = f32(m), f32(p), f32(v)
    res = []
    for sp, flags in ((1, FLAG_WALK_NO_SPLIT), (split, 0)):
        monkeypatch.setenv("BH_WALK_SPLIT", str(sp))
        with engine(n, max_depth=md, reference_compat=False, flags=flags | FLAG_WALK_STATS) as e:
            e.upload(p, v, m)
            e.compute_forces()
            a = e.accelerations()
            st = e.stats()
            e.step(2)
            pos, _ = e.download()
            e.build_tree()                                     # root box from the walk's partials
            root = e.export_tree()[0][0]
        res.append((a, st, pos, root))
    (a1, s1, p1, r1), (a2, s2, p2, r2) = res
    assert (s1.interactions, s1.visits) == (s2.interactions, s2.visits)
    r = rel_err(a2, a1)
    assert np.median(r) < 5e-7 and r.max() < 1e-4
    assert not np.array_equal(a1, a2) or split == 0           # the split really ran (auto may pick 1)
    for pos, root in ((p1, r1), (p2, r2)):
        ex = max(np.ptp(pos[:, 0]), np.ptp(pos[:, 1]))
        np.testing.assert_allclose([root["xmin"], root["xmax"], root["ymin"], root["ymax"]],
                                   [pos[:, 0].min() - 0.1 * ex, pos[:, 0].max() + 0.1 * ex,
                                    pos[:, 1].min() - 0.1 * ex, pos[:, 1].max() + 0.1 * ex], rtol=1e-6)


@pytest.mark.parametrize("precision", [G.Precision.F32, G.Precision.MIXED])
def test_ragged_and_tiny_sizes(precision):
    """n = 0, 1, 2, ... around the wave (64), workgroup (256) and tile (512) boundaries: every launch
    shape of the fp32 pipeline (empty grids, one partial tile, the level-synchronous walk with fewer
    bodies than a wave) against the oracle, through two steps so that the re-ordered state, the walk's
    bounds partials and the integrator are exercised as well."""
    rng = np.random.default_rng(17)
    for n in (0, 1, 2, 3, 5, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1000):
        p = f32(rng.uniform(-0.1, 0.1, (n, 2)))
        v = f32(rng.uniform(-1e-4, 1e-4, (n, 2)))
        m = f32(10.0 ** rng.uniform(-5, -3, n))
        with engine(max(n, 1), precision=precision, max_depth=21, reference_compat=False, flags=FLAG_WALK_STATS) as e:
            e.upload(p, v, m)
            e.compute_forces()
            a = e.accelerations()
            st = e.stats()
            e.step(2)
            p2, v2 = e.download()
        assert a.shape == (n, 2) and p2.shape == (n, 2)
        if n == 0:
            continue
        if n == 1:
            assert np.array_equal(a, np.zeros((1, 2)))
            np.testing.assert_allclose(p2, p + 2 * v, rtol=1e-6)
            continue
        t = O.build_tree(p, m, 0)
        f, ws = O.compute_forces(t, p, m, compat_self_skip=False, with_stats=True)
        ref = f / m[:, None]
        assert st.n_bodies == n and abs(st.interactions - ws.interactions) <= max(2, 1e-3 * ws.interactions)
        r = rel_err(a, ref)
        assert np.median(r) < 1e-5 and r.max() < 5e-3, (n, np.median(r), r.max())
        # two steps of the oracle (uncapped tree, per-body MAC, fp64)
        pos, vel = p.copy(), v.copy()
        for _ in range(2):
            tt = O.build_tree(pos, m, 0)
            _, vel, pos = O.integrate(O.compute_forces(tt, pos, m, compat_self_skip=False), m, vel, pos)
        np.testing.assert_allclose(p2, pos, rtol=0, atol=2e-6 * 0.2)


@pytest.mark.parametrize("theta", [0.2, 0.5, 0.8, 1.2])
@pytest.mark.parametrize("max_depth", [3, 8, 12, 21])
def test_theta_and_depth_cap_matrix_against_the_oracle(theta, max_depth):
    """The reference's two constants (THETA, QUADTREE_MAX_DEPTH; project.cu:60-61) varied: fp32 mode with
    reference_compat = 1 against the oracle's depth-capped tree and walk.  Bodies in multi-occupant cap
    cells feel their own cell's aggregate at tiny distance (the reference's artefact), which amplifies
    fp32 rounding, so they are held to a looser bound."""
    n = 6000
    rng = np.random.default_rng(23)
    p = f32(np.concatenate([rng.normal(0, 0.01, (n // 3, 2)), rng.uniform(-0.1, 0.1, (n - n // 3, 2))]))
    v = f32(np.zeros((n, 2)))
    m = f32(10.0 ** rng.uniform(-2, 1, n))
    t = O.build_tree(p, m, max_depth)
    f, ws = O.compute_forces(t, p, m, theta=theta, with_stats=True)
    ref = f / m[:, None]
    with engine(n, max_depth=max_depth, theta=theta, reference_compat=True, flags=FLAG_WALK_STATS) as e:
        e.upload(p, v, m)
        e.compute_forces()
        a = e.accelerations()
        st = e.stats()
        nodes, _ = e.export_tree()
    assert len(nodes) == len(t)
    assert abs(st.interactions - ws.interactions) <= 2e-3 * ws.interactions + 8     # MAC flips only
    ok = np.linalg.norm(ref, axis=1) > 0
    r = rel_err(a[ok], ref[ok])
    assert np.median(r) <= 5e-6, np.median(r)
    assert np.quantile(r, 0.9) <= 1e-3, np.quantile(r, 0.9)


@pytest.mark.parametrize("precision", ["f32", "mixed"])
def test_random_call_sequences_on_one_context_equal_fresh_contexts(precision):
    """One context, 14 uploads of different sizes in random order (1 ... 300,000 bodies: across the one-bucket sort's 4,096,
    the split walks' launch-size thresholds, the 256-bucket sort's first and later builds), each followed by a random mix of
    bh_step(k) and bh_compute_forces: whatever the previous upload left behind (splitters, sorted copies, bounds records,
    the re-ordering counter), every result is BITWISE that of a fresh context given the same calls."""
    prec = {"f32": G.Precision.F32, "mixed": G.Precision.MIXED}[precision]
    rng = np.random.default_rng(777)
    sizes = [1, 64, 1000, 4096, 4097, 20000, 65536, 70000, 200000, 300000, 130, 8192, 2, 131072]
    rng.shuffle(sizes)
    with engine(300000, max_depth=21, reference_compat=False, precision=prec) as shared:
        for n in sizes:
            m, p, v = IC.make("plummer" if rng.random() < 0.5 else "uniform", int(n), int(rng.integers(1, 100)), quasi_static=True, drift_cells=0.5)
            ops = [(int(rng.integers(0, 2)), int(rng.integers(1, 20))) for _ in range(int(rng.integers(1, 4)))]
            out = []
            for ctx in (shared, None):
                e = ctx if ctx is not None else engine(int(n), max_depth=21, reference_compat=False, precision=prec).__enter__()
                try:
                    e.upload(p, v, m)
                    res = []
                    for op, k in ops:
                        if op == 0:
                            e.step(k)
                        else:
                            res.append(e.compute_forces())
                        res.extend(e.download())
                    out.append(res)
                finally:
                    if ctx is None:
                        e.__exit__(None, None, None)
            assert len(out[0]) == len(out[1])
            for x, y in zip(out[0], out[1]):
                assert np.array_equal(x, y, equal_nan=True), (n, ops)
