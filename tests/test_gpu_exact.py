"""GPU parity, exact mode: libbhgpu (through the C-ABI) vs the reference's own outputs (golden
fixtures) and vs the oracle on the same inputs.  fp64, BIT-EXACT: every comparison is
np.array_equal on the raw doubles -- tree, forces, positions, velocities."""
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import bh_oracle as O  # noqa: E402
import gpu_nbody_simulation_amd as G  # noqa: E402
from gpu_nbody_simulation_amd.engine import FLAG_WALK_STATS  # noqa: E402


def _canon(nodes):
    c = nodes.copy()
    c["child"] = np.where(c["child"] == -1, -1.0, 1.0)
    return c


def _digest(nodes, depth):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(depth, dtype=np.int32).tobytes())
    h.update(np.ascontiguousarray(_canon(nodes)).tobytes())
    return h.hexdigest()


def _same_tree(eng, ref_nodes):
    nodes, depth = eng.export_tree()
    rn, rd = O.canonical_tree(ref_nodes)
    assert len(nodes) == len(rn)
    assert np.array_equal(depth, rd)
    cn = _canon(nodes)
    for f in cn.dtype.names:
        assert np.array_equal(cn[f], rn[f]), f
    # exported child indices are pre-order positions: child k of node i must point forward
    ch = nodes["child"]
    idx = np.arange(len(nodes))[:, None]
    assert np.all((ch == -1) | (ch > idx))


def test_tree_is_the_reference_tree_1024(gold, init1024):
    """buildTree (project.cu:575-591): topology, bounds, COM, mass, occupant -- all fields bitwise."""
    m, p, v = init1024
    g = gold("ref_project_1024")
    with G.BarnesHutEngine(G.BhConfig(capacity=1024)) as e:
        e.upload(p, v, m)
        e.build_tree()
        _same_tree(e, g["tree_0"])
        st = e.stats()
        assert st.n_nodes == 3085 and st.n_internal == 771


def test_forces_and_one_step_bitwise_1024(gold, init1024):
    m, p, v = init1024
    g = gold("ref_project_1024")
    with G.BarnesHutEngine(G.BhConfig(capacity=1024, flags=FLAG_WALK_STATS)) as e:
        e.upload(p, v, m)
        f = e.compute_forces()
        assert np.array_equal(f, g["forces_0"])                       # computeForces, project.cu:593-675
        st = e.stats()
        assert (st.visits, st.interactions) == (150509, 104117)       # the oracle's counts
        assert np.array_equal(e.accelerations(), g["forces_0"] / m[:, None])
        pos0, vel0 = e.download()
        assert np.array_equal(pos0, p) and np.array_equal(vel0, v)    # compute_forces does not advance
        e.step(1)
        pp, vv = e.download()
        assert np.array_equal(pp, g["pos_after_0"]) and np.array_equal(vv, g["vel_after_0"])


@pytest.mark.parametrize("step", [1, 2, 9, 49, 99])
def test_100_steps_bitwise_baseline_config0(gold, init1024, step):
    """BASELINE config[0]: N=1,024, 100 steps on the shipped files -- final positions AND velocities
    equal the reference CPU path's bit for bit, through the ejection and tree collapse of step 1."""
    m, p, v = init1024
    g = gold("ref_project_1024")
    with G.BarnesHutEngine(G.BhConfig(capacity=1024)) as e:
        e.upload(p, v, m)
        e.step(step + 1)
        pp, vv = e.download()
    assert np.array_equal(pp, g[f"pos_after_{step}"])
    assert np.array_equal(vv, g[f"vel_after_{step}"])


def test_collapsed_tree_step1(gold, init1024):
    m, p, v = init1024
    g = gold("ref_project_1024")
    with G.BarnesHutEngine(G.BhConfig(capacity=1024)) as e:
        e.upload(p, v, m)
        e.step(1)
        e.build_tree()
        assert e.stats().n_nodes == 41
        _same_tree(e, g["tree_1"])


@pytest.mark.parametrize("name,steps", [("ref_project_4096", [0, 1, 4, 9]),
                                        ("ref_project_4096_grid", [0, 1, 2, 4, 9, 19])])
def test_4096_cases(gold, name, steps):
    g = gold(name)
    m, p, v = g["mass"], g["pos"], g["vel"]
    with G.BarnesHutEngine(G.BhConfig(capacity=4096)) as e:
        e.upload(p, v, m)
        e.build_tree()
        _same_tree(e, g["tree_0"])
        assert np.array_equal(e.compute_forces(), g["forces_0"])
        done = 0
        for s in steps:
            e.step(s + 1 - done)
            done = s + 1
            pp, vv = e.download()
            assert np.array_equal(pp, g[f"pos_after_{s}"]), s
            assert np.array_equal(vv, g[f"vel_after_{s}"]), s


def test_published_size_40960(gold):
    """N = 40*1024, the size of every number the reference publishes."""
    g = gold("ref_project_40960")
    m, p, v = g["mass"], g["pos"], g["vel"]
    with G.BarnesHutEngine(G.BhConfig(capacity=40960)) as e:
        e.upload(p, v, m)
        assert np.array_equal(e.compute_forces(), g["forces_0"])
        nodes, depth = e.export_tree()
        assert len(nodes) == 97185
        assert _digest(nodes, depth) == str(g["tree_0_sha256"])
        assert np.bincount(depth).tolist() == [1, 4, 16, 64, 256, 784, 3136, 11664, 39992, 41268]


def test_uncapped_tree_matches_main_approach_2(gold):
    """main_approach_2.cpp (no depth cap): max_depth=32, reference_compat off -> `occ == i` only."""
    g = gold("ref_ma2_1000")
    m, p, v = g["mass"], g["pos"], g["vel"]
    with G.BarnesHutEngine(G.BhConfig(capacity=1000, max_depth=32, reference_compat=False)) as e:
        e.upload(p, v, m)
        e.build_tree()
        _same_tree(e, g["tree_0"])
        assert np.array_equal(e.compute_forces(), g["forces_0"])
        done = 0
        for s in (0, 1, 9):
            e.step(s + 1 - done)
            done = s + 1
            pp, vv = e.download()
            assert np.array_equal(pp, g[f"pos_after_{s}"]) and np.array_equal(vv, g[f"vel_after_{s}"])


def test_direct_sum_bounds_the_barnes_hut_error(gold, init1024):
    """main_approach_1.cpp is the physics ground truth (BASELINE config[0] names it): with theta -> 0
    every cell is opened and the walk degenerates to the direct sum, up to summation order."""
    m, p, v = init1024
    g = gold("ref_ma1_1024")
    with G.BarnesHutEngine(G.BhConfig(capacity=1024, max_depth=32, theta=1e-9, reference_compat=False)) as e:
        e.upload(p, v, m)
        f = e.compute_forces()
    ref = g["forces_0"]
    rel = np.linalg.norm(f - ref, axis=1) / np.linalg.norm(ref, axis=1)
    assert rel.max() < 1e-9      # summation order differs; the net force of a body can cancel
    # and at theta = 0.5 the multipole error stays small for almost every body
    with G.BarnesHutEngine(G.BhConfig(capacity=1024, max_depth=32, reference_compat=False)) as e:
        e.upload(p, v, m)
        f = e.compute_forces()
    rel = np.linalg.norm(f - ref, axis=1) / np.linalg.norm(ref, axis=1)
    assert np.median(rel) < 2e-2


@pytest.mark.parametrize("max_depth", [1, 2, 3, 5, 10, 17, 32])
def test_any_depth_cap_matches_the_oracle(max_depth):
    rng = np.random.default_rng(max_depth)
    n = 700
    p = rng.uniform(-1, 1, (n, 2)); v = rng.uniform(-1e-3, 1e-3, (n, 2)); m = 10.0 ** rng.uniform(-2, 1, n)
    p[10] = p[11]                      # coincident pair -> shares every cell down to the cap
    p[20:26] = p[20] + rng.uniform(-1e-9, 1e-9, (6, 2))
    t = O.build_tree(p, m, max_depth)
    with G.BarnesHutEngine(G.BhConfig(capacity=n, max_depth=max_depth)) as e:
        e.upload(p, v, m)
        e.build_tree()
        _same_tree(e, t)
        f = e.compute_forces()
    fo = O.compute_forces(t, p, m)
    # identical including inf/NaN produced by the coincident pair, as in the reference
    assert np.array_equal(f, fo, equal_nan=True)


@pytest.mark.parametrize("n", [0, 1, 2, 3, 63, 64, 65, 257])
def test_tiny_and_ragged_sizes(n):
    rng = np.random.default_rng(100 + n)
    p = rng.uniform(-0.1, 0.1, (n, 2)); v = rng.uniform(-1e-4, 1e-4, (n, 2)); m = 10.0 ** rng.uniform(-2, 1, n)
    with G.BarnesHutEngine(G.BhConfig(capacity=max(n, 1))) as e:
        e.upload(p, v, m)
        e.build_tree()
        _same_tree(e, O.build_tree(p, m, 10))
        e.step(3)
        pp, vv = e.download()
    po, vo = O.run(p, v, m, 3, max_depth=10)
    assert np.array_equal(pp, po) and np.array_equal(vv, vo)


def test_all_bodies_at_one_point():
    """maxDim == 0 -> the 1e-6 pad (project.cu:563-565); everything aggregates in one cap cell."""
    n = 50
    p = np.tile([[0.25, -0.5]], (n, 1)); v = np.zeros((n, 2)); m = np.linspace(1, 2, n)
    t = O.build_tree(p, m, 10)
    with G.BarnesHutEngine(G.BhConfig(capacity=n)) as e:
        e.upload(p, v, m)
        e.build_tree()
        _same_tree(e, t)
        assert np.array_equal(e.compute_forces(), O.compute_forces(t, p, m), equal_nan=True)


def test_quadtree_text_file(gold, init1024, tmp_path):
    """TraverseTreeToFile (project.cu:504-534): byte-identical to the oracle's writer, and identical
    to the REFERENCE's file except the <= 24 lines where it prints out-of-bounds garbage."""
    m, p, v = init1024
    with G.BarnesHutEngine(G.BhConfig(capacity=1024)) as e:
        e.upload(p, v, m)
        e.build_tree()
        e.write_quadtree_file(str(tmp_path / "gpu.txt"))
    O.write_tree_text(O.build_tree(p, m, 10), p, str(tmp_path / "oracle.txt"))
    ours = (tmp_path / "gpu.txt").read_text()
    assert ours == (tmp_path / "oracle.txt").read_text()
    ref = bytes(gold("ref_project_1024")["quadtree_txt_0"]).decode().splitlines()
    mine = ours.splitlines()
    assert len(mine) == len(ref) == 3085
    diff = [(a, b) for a, b in zip(mine, ref) if a != b]
    assert len(diff) <= 24
    for a, b in diff:
        assert int(a.split("occupantIndex=")[1].split()[0]) <= -2
        assert a.split(" occupantPos=")[0] == b.split(" occupantPos=")[0]
    from gpu_nbody_simulation_amd.textio import parse_quadtree_file
    e0 = parse_quadtree_file(str(tmp_path / "gpu.txt"))[0]
    assert e0 == (0, -0.119497, 0.119541, -0.119883, 0.11995, 1568.43, [(-1, 0.000603463, -0.00254328)])


def test_body_order_is_the_callers(init1024):
    """Shuffling the input order permutes the outputs and changes nothing else (the tree is order
    independent; inside a cap cell the fold follows body order, so only cap-cell-free inputs are
    bitwise permutation invariant -- use a well separated set)."""
    rng = np.random.default_rng(5)
    n = 512
    g = int(np.ceil(np.sqrt(n)))
    ij = np.stack(np.meshgrid(np.arange(g), np.arange(g), indexing="ij"), -1).reshape(-1, 2)[:n]
    p = -0.1 + (ij + 0.5 + rng.uniform(-0.3, 0.3, (n, 2))) * (0.2 / g)
    v = rng.uniform(-1e-4, 1e-4, (n, 2)); m = 10.0 ** rng.uniform(-2, 1, n)
    perm = rng.permutation(n)
    with G.BarnesHutEngine(G.BhConfig(capacity=n)) as e:
        e.upload(p, v, m); e.step(5); a, b = e.download()
        e.upload(p[perm], v[perm], m[perm]); e.step(5); a2, b2 = e.download()
    assert np.array_equal(a[perm], a2) and np.array_equal(b[perm], b2)


@pytest.mark.parametrize("theta", [0.1, 0.3, 0.7, 1.0, 2.5])
@pytest.mark.parametrize("compat", [True, False])
def test_any_theta_matches_the_oracle_bitwise(theta, compat):
    """THETA (project.cu:60) varied, both self-skip rules (project.cu:646 / main_approach_2.cpp): forces
    and three steps bitwise equal to the oracle; G and dt varied along."""
    rng = np.random.default_rng(int(theta * 100) + compat)
    n = 3000
    p = np.concatenate([rng.normal(0, 0.01, (n // 2, 2)), rng.uniform(-0.1, 0.1, (n - n // 2, 2))])
    v = rng.uniform(-1e-4, 1e-4, (n, 2)); m = 10.0 ** rng.uniform(-2, 1, n)
    Gc, dt = 6.67e-11 * 3.0, 0.25
    t = O.build_tree(p, m, 10)
    fo = O.compute_forces(t, p, m, theta=theta, G=Gc, compat_self_skip=compat)
    with G.BarnesHutEngine(G.BhConfig(capacity=n, max_depth=10, theta=theta, G=Gc, dt=dt, reference_compat=compat)) as e:
        e.upload(p, v, m)
        f = e.compute_forces()
        e.step(3)
        pp, vv = e.download()
    assert np.array_equal(f, fo, equal_nan=True)
    pos, vel = p.copy(), v.copy()
    for _ in range(3):
        tt = O.build_tree(pos, m, 10)
        _, vel, pos = O.integrate(O.compute_forces(tt, pos, m, theta=theta, G=Gc, compat_self_skip=compat), m, vel, pos, dt=dt)
    # (a body inside a multi-occupant cap cell can sit exactly on the aggregate: inf * 0 = NaN on both sides)
    assert np.array_equal(pp, pos, equal_nan=True) and np.array_equal(vv, vel, equal_nan=True)


@pytest.mark.parametrize("precision", [G.Precision.F64_EXACT, G.Precision.F64])
@pytest.mark.parametrize("n", [1000, 5000, 40960])
def test_bodies_per_wavefront(gold, precision, n):
    """Round 4: launches of few bodies give every wavefront of the fp64 walks fewer than 64 bodies (1 at N <= 2,048 / 4,096) so that
    its walk -- one dependent chain over the union of its bodies' walks -- is short (config 1: 0.30 -> 0.10 ms).  In the BIT-EXACT
    mode every lane adds its own terms in the reference's order whoever shares its wave: forces, counters and a 3-step trajectory
    are bitwise the same for 1, 4, 16 and 64 bodies per wave and for the engine's own choice -- and the reference's.  The
    THROUGHPUT mode adds a lane's terms in the order its wave meets them: same counters, forces within 1e-13 of each other, and
    bitwise equal again under BH_FLAG_WALK_NO_SPLIT, which pins 64 bodies per wave."""
    from gpu_nbody_simulation_amd.engine import FLAG_WALK_NO_SPLIT
    g = gold("ref_project_40960")
    m, p, v = g["mass"][:n], g["pos"][:n], g["vel"][:n]
    exact = precision == G.Precision.F64_EXACT
    out, pinned = [], []
    try:
        for bpw in ("0", "1", "4", "16", "64"):
            os.environ["BH_EXACT_BPW"] = bpw
            for flags, dst in ((FLAG_WALK_STATS, out), (FLAG_WALK_STATS | FLAG_WALK_NO_SPLIT, pinned)):
                with G.BarnesHutEngine(G.BhConfig(capacity=n, precision=precision, flags=flags)) as e:
                    e.upload(p, v, m)
                    f = e.compute_forces()
                    st = e.stats()
                    e.step(3)
                    dst.append((f, (st.visits, st.interactions)) + e.download())
    finally:
        os.environ.pop("BH_EXACT_BPW", None)
    same = lambda a, b: all(np.array_equal(x, y, equal_nan=True) for x, y in ((a[0], b[0]), (a[2], b[2]), (a[3], b[3]))) and a[1] == b[1]
    for o in out[1:]:
        if exact:
            assert same(o, out[0])
        else:
            ok = np.isfinite(out[0][0]).all(axis=1)
            r = np.linalg.norm(o[0][ok] - out[0][0][ok], axis=1) / np.linalg.norm(out[0][0][ok], axis=1)
            assert o[1] == out[0][1] and r.max() <= 1e-13
    if not exact:
        for o in pinned[1:]:
            assert same(o, pinned[0])
    if exact and n == 40960:
        assert np.array_equal(out[0][0], g["forces_0"])                              # (and they are the reference's)


def _threshold_cases():
    rng = np.random.default_rng(11)

    def clumped(n):
        p = np.concatenate([rng.normal(0, 1e-3, (n // 2, 2)), rng.uniform(-1, 1, (n - n // 2, 2))])
        m, v = rng.uniform(0.1, 0.5, n), rng.uniform(-1e-9, 1e-9, (n, 2))
        m[::7] = 1e-16                                            # below the reference's 1e-15 cut-off
        return m, p, v

    def lattice(n):                                               # dx = 0 / dy = 0 exactly between many pairs
        k = int(np.sqrt(n))
        g = np.stack(np.meshgrid(np.arange(k), np.arange(k)), -1).reshape(-1, 2).astype(np.float64) / 64.0
        return np.full(len(g), 0.25), g, np.zeros_like(g)

    cases = {}
    m, p, v = clumped(30000); cases["clumped"] = (m, p, v, dict(max_depth=8))
    m, p, v = clumped(30000); cases["clumped_deep"] = (m, p, v, dict(max_depth=32, reference_compat=False))
    m, p, v = lattice(16384); cases["lattice"] = (m, p, v, dict(max_depth=12))
    # operand ranges where IEEE division / sqrt need their range handling: the walk must leave its short sequences
    m, p, v = clumped(20000); cases["huge_coordinates"] = (m, p * 1e130, v, dict(max_depth=21))      # d2 ~ 1e260 > 2^400
    m, p, v = clumped(20000); cases["tiny_coordinates"] = (m, p * 1e-140, v, dict(max_depth=21))     # d2 ~ 1e-280 < 2^-400
    m, p, v = clumped(20000); cases["tiny_G"] = (m, p, v, dict(max_depth=21, G=1e-130))              # G m_i < 2^-150
    m, p, v = clumped(20000); cases["huge_masses"] = (m * 1e80, p, v * 0, dict(max_depth=21, dt=1e-200))   # node masses > 2^150
    m, p, v = clumped(20000); p[5] = (1e200, -1e200); p[77] = (3e-300, 1.0)                          # d2 = inf from one body; a denormal-ish dx
    cases["one_body_at_1e200"] = (m, p, v, dict(max_depth=21))
    m, p, v = clumped(20000); m[::3] *= 1e-10; m[1::3] *= 1e60                                        # lanes of one wave on both sides of the range test
    cases["mixed_mass_ranges"] = (m, p, v, dict(max_depth=21))
    for th in (1e-9, 0.01, 1.7, 40.0):
        m, p, v = clumped(8000); cases[f"theta_{th}"] = (m, p, v, dict(max_depth=14, theta=th))
    return cases


_THR_CASES = _threshold_cases()


@pytest.mark.parametrize("name", sorted(_THR_CASES))
def test_threshold_walk_equals_the_walk_written_as_the_reference_writes_it(name):
    """Round 4: the bit-exact walk decides acceptance by comparing d2 with the node's EXACT threshold (the smallest double
    whose sqrt, + 1e-15 and division give size / d < theta: exact_walk_threshold, csrc/bh_tree.hpp) and runs sqrt and the
    three divisions of an accepted term through the compiler's own instruction sequences without their range handling
    whenever every taking lane's operands are inside the range where that handling is the identity
    (csrc/bh_walk_exact.hpp).  BH_FLAG_WALK_PORTABLE runs the walk as the reference writes it -- sqrt, size / d < theta,
    three divisions -- on a tree whose nodes carry their sizes.  Forces, counters and the state after 3 steps are BITWISE
    equal on ordinary inputs and on inputs that leave every range (d2 = inf and 1e-280, G m = 1e-150, exact zeros in dx)."""
    from gpu_nbody_simulation_amd.engine import FLAG_WALK_PORTABLE
    m, p, v, kw = _THR_CASES[name]
    n = len(m)
    res = []
    for flags in (FLAG_WALK_STATS, FLAG_WALK_STATS | FLAG_WALK_PORTABLE, 0):
        with G.BarnesHutEngine(G.BhConfig(capacity=n, flags=flags, **kw)) as e:
            e.upload(p, v, m)
            f = e.compute_forces()
            st = e.stats()
            e.step(3)
            res.append((f, (st.visits, st.interactions) if flags else None) + e.download())
    assert res[0][1] == res[1][1] and res[0][1][1] > 0
    for other in res[1:]:
        for x, y in zip(res[0], other):
            if x is not None and y is not None and not isinstance(x, tuple):
                assert np.array_equal(x, y, equal_nan=True)
    assert np.isfinite(res[0][0]).mean() > 0.9                    # (not a comparison of NaNs)


def test_threshold_walk_full_size():
    """... and on all 1,048,576 bodies of config 3 (Plummer, theta 0.5, uncapped)."""
    from gpu_nbody_simulation_amd.engine import FLAG_WALK_PORTABLE
    from gpu_nbody_simulation_amd import initial_conditions as IC
    n = 1 << 20
    m, p, v = IC.make("plummer", n, 1, quasi_static=True)
    res = []
    for flags in (0, FLAG_WALK_PORTABLE):
        with G.BarnesHutEngine(G.BhConfig(capacity=n, max_depth=21, flags=flags)) as e:
            e.upload(p, v, m)
            f = e.compute_forces()
            e.step(2)
            res.append((f,) + e.download())
    assert np.isfinite(res[0][0]).all() and np.abs(res[0][0]).max() > 0
    for x, y in zip(res[0], res[1]):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("n,theta,md,compat", [(1500, 0.5, 10, True), (1500, 1e-6, 10, True), (4096, 0.05, 21, False),
                                                (4097, 0.5, 21, True), (6000, 0.02, 12, True), (8192, 0.5, 32, False),
                                                (3, 0.5, 10, True), (257, 0.5, 2, True)])
def test_one_wavefront_per_body_walk_equals_the_cooperative_walk(monkeypatch, n, theta, md, compat):
    """Round 4: launches of up to 8,192 bodies walk the tree with ONE WAVEFRONT PER BODY, breadth-first (walk_exact_bfs_kernel):
    every lane decides one queued node, the accepted terms carry the DFS key of their node and are added in ascending key
    order -- the reference's order of additions.  BITWISE the same forces and trajectory as the cooperative walk (an explicit
    BH_EXACT_BPW keeps it) and as the walk written as the reference writes it; theta -> 0 and dense depth-cap trees overflow
    the 256 / 384-term list and the node queue, and the walk starts again through the assembly loop -- the same bits."""
    from gpu_nbody_simulation_amd.engine import FLAG_WALK_PORTABLE
    rng = np.random.default_rng(n)
    p = np.concatenate([rng.normal(0, 3e-2, (n // 2, 2)), rng.uniform(-1, 1, (n - n // 2, 2))])
    m, v = rng.uniform(0.1, 0.5, n), rng.uniform(-1e-7, 1e-7, (n, 2))
    m[::9] = 1e-16
    res = []
    for bpw, flags in ((None, 0), ("64", 0), ("1", 0), (None, FLAG_WALK_PORTABLE)):
        if bpw is None:
            monkeypatch.delenv("BH_EXACT_BPW", raising=False)
        else:
            monkeypatch.setenv("BH_EXACT_BPW", bpw)
        with G.BarnesHutEngine(G.BhConfig(capacity=n, theta=theta, max_depth=md, reference_compat=compat, flags=flags)) as e:
            e.upload(p, v, m)
            f = e.compute_forces()
            e.step(3)
            res.append((f,) + e.download())
    # (a body of a multi-occupant depth-cap cell can sit exactly on the aggregate: inf * 0 = NaN in the reference too)
    assert np.isfinite(res[0][0]).mean() > 0.9 and np.nanmax(np.abs(res[0][0])) > 0
    for other in res[1:]:
        for x, y in zip(res[0], other):
            assert np.array_equal(x, y, equal_nan=True)


def test_one_wavefront_per_body_walk_on_random_small_trees(monkeypatch):
    """120 seeded random cases: 1 to 700 bodies, clustered with exact duplicates (many occupants per depth-cap cell), zero and
    tiny masses, depth caps 1 to 32, theta from 1e-3 to 3, both occupant rules -- the breadth-first walk of small launches
    against the cooperative walk, forces and two steps BITWISE (NaNs of the reference's own inf * 0 included)."""
    rng = np.random.default_rng(2024)
    for case in range(120):
        n = int(rng.integers(1, 700))
        md = int(rng.choice([1, 2, 3, 5, 8, 10, 16, 21, 32]))
        theta = float(10.0 ** rng.uniform(-3, 0.5))
        compat = bool(rng.integers(0, 2))
        centres = rng.uniform(-1, 1, (int(rng.integers(1, 6)), 2))
        p = centres[rng.integers(0, len(centres), n)] + rng.normal(0, 10.0 ** rng.uniform(-6, -1), (n, 2))
        if n > 4:
            p[rng.integers(0, n, n // 5)] = p[rng.integers(0, n, n // 5)]          # exact duplicates
        m = 10.0 ** rng.uniform(-3, 1, n)
        m[rng.random(n) < 0.1] = 0.0
        m[rng.random(n) < 0.05] = 1e-16
        v = rng.normal(0, 1e-6, (n, 2))
        res = []
        for bpw in (None, "64"):
            if bpw is None:
                monkeypatch.delenv("BH_EXACT_BPW", raising=False)
            else:
                monkeypatch.setenv("BH_EXACT_BPW", bpw)
            # (exact duplicates under a deep cap are chains of 30 nested cells: room for them)
            with G.BarnesHutEngine(G.BhConfig(capacity=n, theta=theta, max_depth=md, reference_compat=compat,
                                              node_capacity=140 * n + 4096)) as e:
                e.upload(p, v, m)
                f = e.compute_forces()
                e.step(2)
                res.append((f,) + e.download())
        for x, y in zip(res[0], res[1]):
            assert np.array_equal(x, y, equal_nan=True), (case, n, md, theta, compat)


@pytest.mark.parametrize("bpw", [None, "64", "16"])
@pytest.mark.parametrize("precision", [G.Precision.F64_EXACT, G.Precision.F64])
def test_a_body_gone_nan_does_not_take_the_root_box_along(monkeypatch, precision, bpw):
    """A massless body's acceleration is 0 / 0 (project.cu:827): its position is NaN from the first step on.  The reference
    folds the root box with std::min / std::max from +-inf (project.cu:544-551), which a NaN never wins, so everybody else
    carries on -- the oracle says so.  The walks' epilogues reduce the new positions over the wave: a lane whose OWN value was
    NaN kept it, and if that lane was the one the reduction ends in, the next root box was NaN and every body with it (found
    by the random test above, fixed in wave_min / wave_max).  Three steps against the oracle, NaNs in the same places."""
    rng = np.random.default_rng(8)
    n = 700
    p, v = rng.uniform(-1, 1, (n, 2)), rng.normal(0, 1e-6, (n, 2))
    m = rng.uniform(0.1, 1.0, n)
    m[63] = 0.0; m[64 * 5 + 15] = 0.0; m[n - 1] = 0.0; m[128] = 0.0           # the last lane of a wave, of a 16-body wave, ...
    if bpw is None:
        monkeypatch.delenv("BH_EXACT_BPW", raising=False)
    else:
        monkeypatch.setenv("BH_EXACT_BPW", bpw)
    pos, vel = p.copy(), v.copy()
    with G.BarnesHutEngine(G.BhConfig(capacity=n, max_depth=10, precision=precision)) as e:
        e.upload(p, v, m)
        for step in range(3):
            t = O.build_tree(pos, m, 10)
            _, vel, pos = O.integrate(O.compute_forces(t, pos, m), m, vel, pos)
            e.step(1)
            pp, vv = e.download()
            assert int(np.isnan(pos).any(axis=1).sum()) == 4
            ok = ~np.isnan(pos).any(axis=1)
            if precision == G.Precision.F64_EXACT:
                assert np.array_equal(pp, pos, equal_nan=True) and np.array_equal(vv, vel, equal_nan=True)
            else:
                # (the throughput walk sums accelerations, not forces: its massless bodies move on finite orbits; everybody
                #  else agrees with the oracle, whose massless bodies -- NaN, hence invisible to its tree -- pull nobody)
                assert np.isfinite(pp).all()
                assert np.abs(pp[ok] - pos[ok]).max() <= 1e-11 * np.ptp(pos[ok], axis=0).max()


def test_random_small_systems_against_the_oracle():
    """80 seeded random systems (1 to 500 bodies; clusters, exact duplicates, masses from 1e-16 to 100, depth caps 1 to 32,
    theta 1e-2 to 2, both occupant rules, G and dt varied): tree, forces and three steps BITWISE equal to the oracle's --
    NaNs (bodies on a depth-cap aggregate) in the same places with everybody else carrying on.
    Masses are POSITIVE: QuadInsert takes a leaf whose mass is 0.0 for empty (project.cu:395-397), so a massless body is
    overwritten by a later arrival but subdivides an earlier one -- an order-dependent tree this sort-based build does not
    reproduce (the oracle does, which is how this test found out)."""
    rng = np.random.default_rng(77)
    for case in range(80):
        n = int(rng.integers(1, 500))
        md = int(rng.choice([1, 2, 4, 7, 10, 15, 21, 32]))
        theta = float(10.0 ** rng.uniform(-2, 0.3))
        compat = bool(rng.integers(0, 2))
        Gc, dt = 6.67e-11 * float(10.0 ** rng.uniform(-2, 2)), float(rng.choice([1.0, 0.25, 3.0]))
        centres = rng.uniform(-1, 1, (int(rng.integers(1, 5)), 2))
        p = centres[rng.integers(0, len(centres), n)] + rng.normal(0, 10.0 ** rng.uniform(-5, -1), (n, 2))
        if n > 4:
            p[rng.integers(0, n, n // 6)] = p[rng.integers(0, n, n // 6)]
        m = 10.0 ** rng.uniform(-4, 2, n)
        m[rng.random(n) < 0.08] = 1e-16
        v = rng.normal(0, 1e-5, (n, 2))
        with G.BarnesHutEngine(G.BhConfig(capacity=n, theta=theta, G=Gc, dt=dt, max_depth=md, reference_compat=compat,
                                          node_capacity=140 * n + 4096)) as e:
            e.upload(p, v, m)
            f = e.compute_forces()
            t = O.build_tree(p, m, md)
            _same_tree(e, t)
            fo = O.compute_forces(t, p, m, theta=theta, G=Gc, compat_self_skip=compat)
            assert np.array_equal(f, fo, equal_nan=True), (case, n, md, theta, compat)
            pos, vel = p.copy(), v.copy()
            for step in range(3):
                tt = O.build_tree(pos, m, md)
                _, vel, pos = O.integrate(O.compute_forces(tt, pos, m, theta=theta, G=Gc, compat_self_skip=compat), m, vel, pos, dt=dt)
                e.step(1)
                pp, vv = e.download()
                assert np.array_equal(pp, pos, equal_nan=True) and np.array_equal(vv, vel, equal_nan=True), (case, step, n, md, theta, compat)


def test_random_call_sequences_on_one_context_against_the_oracle():
    """One context, 14 uploads of different sizes in random order (1 ... 20,000 bodies: across the one-bucket sort's 4,096,
    the one-wavefront-per-body walk's 12,288 and back), each followed by a random mix of bh_step(k), bh_compute_forces and
    tree exports: whatever the previous upload left behind (splitters, bounds records, their count, the sorted order), the
    state after every call is the oracle's, BITWISE."""
    from gpu_nbody_simulation_amd import initial_conditions as IC
    rng = np.random.default_rng(4242)
    sizes = [1, 50, 1000, 4096, 4097, 5000, 12288, 12289, 20000, 700, 8192, 3, 16384, 1024]
    rng.shuffle(sizes)
    with G.BarnesHutEngine(G.BhConfig(capacity=20000, max_depth=10, node_capacity=200000)) as e:
        for n in sizes:
            m, p, v = IC.make("plummer" if rng.random() < 0.5 else "uniform", int(n), int(rng.integers(1, 100)), quasi_static=True)
            pos, vel = p.copy(), v.copy()
            e.upload(p, v, m)
            for _ in range(int(rng.integers(1, 4))):
                op = rng.integers(0, 3)
                if op == 0:
                    k = int(rng.integers(1, 4))
                    e.step(k)
                    for _s in range(k):
                        t = O.build_tree(pos, m, 10)
                        _, vel, pos = O.integrate(O.compute_forces(t, pos, m), m, vel, pos)
                elif op == 1:
                    f = e.compute_forces()
                    assert np.array_equal(f, O.compute_forces(O.build_tree(pos, m, 10), pos, m), equal_nan=True), n
                else:
                    e.build_tree()
                    _same_tree(e, O.build_tree(pos, m, 10))
                pp, vv = e.download()
                assert np.array_equal(pp, pos, equal_nan=True) and np.array_equal(vv, vel, equal_nan=True), (n, op)
