"""Pins the CPU oracle (oracle/bh_oracle.c) to the REFERENCE's own compiled code.

Every array in tests/golden/ref_*.npz was written by the reference's functions
(project.cu CPU path, main_approach_1.cpp, main_approach_2.cpp) through oracle/ref_driver.cpp
in the development container (scripts/make_golden.py).  The oracle must reproduce them
bit for bit in fp64 -- tree (including the reference's node numbering), forces, positions and
velocities -- so every comparison here is np.array_equal, not allclose.
"""
import hashlib
import os

import numpy as np
import pytest

from oracle import bh_oracle as O


def _digest(nodes):
    can, depth = O.canonical_tree(nodes)
    h = hashlib.sha256()
    h.update(depth.tobytes())
    h.update(can.tobytes())
    return h.hexdigest(), np.bincount(depth)


def _same_tree(a, b):
    return np.array_equal(a.view(np.float64), b.view(np.float64))


def test_project_1024_step0_tree_forces_state(gold, init1024):
    """project.cu:575-591 (tree), :593-675 (forces), :795-817 (integrator) at BASELINE config[0]."""
    g = gold("ref_project_1024")
    m, p, v = init1024
    t = O.build_tree(p, m, 10)
    assert len(t) == 3085 == int(g["tree_0_n_nodes"])          # SURVEY 8(c) known answer
    assert _same_tree(t, g["tree_0"])
    f, st = O.compute_forces(t, p, m, with_stats=True)
    assert np.array_equal(f, g["forces_0"])
    assert st.interactions == 104117 and st.max_stack == 20     # 101.7/body, stack 20
    _, v1, p1 = O.integrate(f, m, v, p)
    assert np.array_equal(p1, g["pos_after_0"]) and np.array_equal(v1, g["vel_after_0"])
    a = np.hypot(f[:, 0], f[:, 1]) / m
    assert a.argmax() == 397 and abs(a.max() - 91.48) < 0.01


def test_project_1024_depth_histogram(gold):
    g = gold("ref_project_1024")
    assert g["tree_0_depth_hist"].tolist() == [1, 4, 16, 64, 252, 760, 1168, 580, 184, 56]


@pytest.mark.parametrize("step", [1, 2, 9, 49, 99])
def test_project_1024_multistep_bitwise(gold, init1024, step):
    """100 steps of runSimulationCpu (project.cu:883-910), degenerate collapse included."""
    g = gold("ref_project_1024")
    m, p, v = init1024
    pp, vv = O.run(p, v, m, step + 1, max_depth=10)
    assert np.array_equal(pp, g[f"pos_after_{step}"])
    assert np.array_equal(vv, g[f"vel_after_{step}"])


def test_project_1024_collapsed_tree_step1(gold, init1024):
    g = gold("ref_project_1024")
    m, p, v = init1024
    p1 = g["pos_after_0"]
    t = O.build_tree(p1, m, 10)
    assert len(t) == 41 and _same_tree(t, g["tree_1"])          # 3085 -> 41


@pytest.mark.parametrize("name,steps", [("ref_project_4096", [0, 1, 4, 9]),
                                        ("ref_project_4096_grid", [0, 1, 2, 4, 9, 19])])
def test_project_4096(gold, name, steps):
    g = gold(name)
    m, p, v = g["mass"], g["pos"], g["vel"]
    t = O.build_tree(p, m, 10)
    assert _same_tree(t, g["tree_0"])
    assert np.array_equal(O.compute_forces(t, p, m), g["forces_0"])
    for s in steps:
        pp, vv = O.run(p, v, m, s + 1, max_depth=10)
        assert np.array_equal(pp, g[f"pos_after_{s}"]), s
        assert np.array_equal(vv, g[f"vel_after_{s}"]), s
        if s > 0:
            ps = g[f"pos_after_{s - 1}"] if f"pos_after_{s - 1}" in g else None
            if ps is not None:
                dig, hist = _digest(O.build_tree(ps, m, 10))
                assert dig == str(g[f"tree_{s}_sha256"])


def test_grid_case_has_no_shared_cap_cells(gold):
    """The synthetic case is the encounter-free one: every cap cell holds at most one body."""
    g = gold("ref_project_4096_grid")
    t = g["tree_0"]
    leaf = t["child"][:, 0] == -1
    assert not np.any(leaf & (t["particle"] == -1) & (t["mass"] > 0))


def test_project_40960_published_size(gold):
    """N = 40*1024, the size every published number of the reference is quoted on."""
    g = gold("ref_project_40960")
    m, p = g["mass"], g["pos"]
    t = O.build_tree(p, m, 10)
    assert len(t) == 97185 == int(g["tree_0_n_nodes"])
    dig, hist = _digest(t)
    assert dig == str(g["tree_0_sha256"])
    assert hist.tolist() == [1, 4, 16, 64, 256, 784, 3136, 11664, 39992, 41268]
    f, st = O.compute_forces(t, p, m, with_stats=True)
    assert np.array_equal(f, g["forces_0"])
    assert st.max_stack == 24 and round(st.interactions / 40960, 1) == 199.5


def test_ma2_uncapped(gold):
    """main_approach_2.cpp:73-175, 261-343: uncapped tree, `occ == i` self skip."""
    g = gold("ref_ma2_1000")
    m, p, v = g["mass"], g["pos"], g["vel"]
    t = O.build_tree(p, m, 0)
    assert _same_tree(t, g["tree_0"])
    f = O.compute_forces(t, p, m, compat_self_skip=False)
    assert np.array_equal(f, g["forces_0"])
    for s in (0, 1, 9):
        pp, vv = O.run(p, v, m, s + 1, max_depth=0)
        assert np.array_equal(pp, g[f"pos_after_{s}"]) and np.array_equal(vv, g[f"vel_after_{s}"])


def test_ma1_direct_sum(gold, init1024):
    """main_approach_1.cpp:53-75, 139-148 with n set to 1,024 (BASELINE config[0])."""
    g = gold("ref_ma1_1024")
    m, p, v = init1024
    assert np.array_equal(O.direct_forces(p, m), g["forces_0"])
    for s in (0, 9, 99):
        pp, vv = O.run(p, v, m, s + 1, direct=True)
        assert np.array_equal(pp, g[f"pos_after_{s}"]) and np.array_equal(vv, g[f"vel_after_{s}"])


def test_tree_text_dump_matches_reference_file(gold, init1024, tmp_path):
    """project.cu:504-534.  All lines equal the reference's file except the 24 lines whose
    occupant index is <= -2, where the reference prints out-of-bounds garbage."""
    g = gold("ref_project_1024")
    m, p, v = init1024
    t = O.build_tree(p, m, 10)
    out = tmp_path / "quadtree_init_cpu.txt"
    assert O.write_tree_text(t, p, str(out)) == 3085
    ours = out.read_text().splitlines()
    ref = bytes(g["quadtree_txt_0"]).decode().splitlines()
    assert len(ours) == len(ref) == 3085
    assert ref[0] == ("0 -0.119497 0.119541 -0.119883 0.11995 1568.43 occupantIndex=-1 "
                      "occupantPos=(0.000603463,-0.00254328)")
    bad = 0
    for a, b in zip(ours, ref):
        if a != b:
            bad += 1
            idx = int(a.split("occupantIndex=")[1].split()[0])
            assert idx <= -2
            assert a.split(" occupantIndex=")[0] == b.split(" occupantIndex=")[0]
            body = -(idx + 2)
            assert a.endswith("occupantPos=(%g,%g)" % (p[body, 0], p[body, 1]))
    assert bad <= 24
    n_le_m2 = sum(1 for a in ours if "occupantIndex=-" in a and int(a.split("occupantIndex=")[1].split()[0]) <= -2)
    assert n_le_m2 == 24


def test_root_bounds_degenerate():
    b = O.root_bounds(np.array([[0.25, -0.5]]))
    assert b.tolist() == [0.25 - 1e-6, 0.25 + 1e-6, -0.5 - 1e-6, -0.5 + 1e-6]   # project.cu:563-565


def test_empty_and_single_body():
    t = O.build_tree(np.zeros((0, 2)), np.zeros(0), 10)
    assert len(t) == 1 and t["mass"][0] == 0
    t = O.build_tree(np.array([[0.1, 0.2]]), np.array([3.0]), 10)
    assert len(t) == 1 and t["particle"][0] == 0 and t["mass"][0] == 3.0
    f = O.compute_forces(t, np.array([[0.1, 0.2]]), np.array([3.0]))
    assert np.array_equal(f, np.zeros((1, 2)))


def test_diagnostic_walk_reproduces_the_pinned_walk(gold):
    """bho_compute_forces_diag (the fp32 parity tests' per-body diagnostics) restates the pinned walk: forces
    bit for bit, per-body counts adding up to the walk's own counter -- on the reference's shipped bodies
    (depth cap 10, the reference's self skip) and on the uncapped tree; any thread split gives the same arrays."""
    g = gold("ref_project_40960")
    m, p = g["mass"], g["pos"]
    for md, compat in ((10, True), (0, False)):
        t = O.build_tree(p, m, md)
        f, st = O.compute_forces(t, p, m, compat_self_skip=compat, with_stats=True)
        d1 = O.compute_forces_diag(t, p, m, compat_self_skip=compat, threads=1)
        d4 = O.compute_forces_diag(t, p, m, compat_self_skip=compat, threads=4)
        assert np.array_equal(f, d1.forces) and np.array_equal(f, d4.forces)
        assert int(d1.counts.sum()) == st.interactions and np.array_equal(d1.counts, d4.counts)
        for a, b in ((d1.abs_sum, d4.abs_sum), (d1.coord, d4.coord), (d1.flip, d4.flip)):
            assert np.array_equal(a, b)
        assert (d1.abs_sum >= np.linalg.norm(f, axis=1) * (1 - 1e-12)).all()      # triangle inequality
        assert 0 < (d1.flip > 0).mean() < 0.01                                    # borderline cells are rare
    if md == 10:
        assert np.array_equal(f, g["forces_0"])                                   # (and the walk itself is the golden one)
