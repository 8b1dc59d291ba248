"""C-ABI behaviour on a GPU: error codes and messages, call-order errors, stats, file errors."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import gpu_nbody_simulation_amd as G  # noqa: E402
from gpu_nbody_simulation_amd import _lib  # noqa: E402


def _bodies(n, seed=0):
    r = np.random.default_rng(seed)
    return 10.0 ** r.uniform(-2, 1, n), r.uniform(-0.1, 0.1, (n, 2)), r.uniform(-1e-4, 1e-4, (n, 2))


def test_upload_more_than_capacity_is_the_references_out_of_range():
    m, p, v = _bodies(10)
    with G.BarnesHutEngine(G.BhConfig(capacity=8)) as e:
        with pytest.raises(G.BhError) as ei:
            e.upload(p, v, m)
        assert ei.value.code == -1 and "Requested number of bodies exceeds N_BODIES." in str(ei.value)


def test_call_order_errors():
    with G.BarnesHutEngine(G.BhConfig(capacity=8)) as e:
        for fn in (e.step, e.build_tree, e.compute_forces):
            with pytest.raises(G.BhError) as ei:
                fn()
            assert ei.value.code == -5
        with pytest.raises(G.BhError):
            e.export_tree()


def test_export_buffer_too_small_reports_needed_size():
    m, p, v = _bodies(64)
    lib = _lib.load()
    with G.BarnesHutEngine(G.BhConfig(capacity=64)) as e:
        e.upload(p, v, m)
        e.build_tree()
        n = C.c_int64(0)
        rc = lib.bh_export_tree(e._h, None, None, 0, C.byref(n))
        assert rc == -4 and n.value == e.stats().n_nodes > 1


def test_node_capacity_overflow_is_reported_not_fatal():
    m, p, v = _bodies(4096)
    with G.BarnesHutEngine(G.BhConfig(capacity=4096, node_capacity=101)) as e:
        e.upload(p, v, m)
        with pytest.raises(G.BhError) as ei:
            e.build_tree()
        assert ei.value.code == -4
        with pytest.raises(G.BhError):
            e.compute_forces()


def test_write_file_error():
    m, p, v = _bodies(16)
    with G.BarnesHutEngine(G.BhConfig(capacity=16)) as e:
        e.upload(p, v, m)
        e.build_tree()
        with pytest.raises(G.BhError) as ei:
            e.write_quadtree_file("/nonexistent_dir/x.txt")
        assert ei.value.code == -6


def test_stats_and_timers():
    m, p, v = _bodies(20000)
    with G.BarnesHutEngine(G.BhConfig(capacity=20000, precision=G.Precision.F32, max_depth=16)) as e:
        e.upload(p, v, m)
        e.step(4)
        st = e.stats()
        assert st.n_bodies == 20000 and st.steps_done == 4 and st.n_nodes == 1 + 4 * st.n_internal
        assert 0 < st.walk_ms < st.last_step_ms * 1.01 and st.build_ms > 0 and st.device_bytes > 0
        # per kernel group of the last timed step: the groups add up to the build
        groups = st.keys_ms + st.sort_ms + st.scan_ms + st.nodes_ms
        assert min(st.keys_ms, st.sort_ms, st.scan_ms, st.nodes_ms) > 0 and abs(groups - st.build_ms) < 0.02 * st.build_ms + 1e-3
        # algorithmic bytes: the 4th build sorted with the bucket sort (46 B per body instead of 5 x 24)
        # (these bodies are heavy: close encounters blow the root box up every step, the keys collapse into a few
        # cells and a bucket of the sort may outgrow its LDS buffer -- counted, never wrong)
        assert st.build_bytes == 20000 * (16 + 46 + 40 + 47 + 117) and st.sort_spill_buckets >= 0
        assert st.walk_bytes == 0                                   # needs BH_FLAG_WALK_STATS
        e.upload(p, v, m)
        assert e.stats().steps_done == 0


def test_reupload_smaller_and_reuse_context():
    m, p, v = _bodies(1000, 1)
    with G.BarnesHutEngine(G.BhConfig(capacity=1000)) as e:
        e.upload(p, v, m); e.step(2); a = e.download()
        e.upload(p[:100], v[:100], m[:100]); e.step(2)
        e.upload(p, v, m); e.step(2); b = e.download()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("precision", [G.Precision.F32, G.Precision.MIXED])
def test_periodic_reordering_of_the_state_is_invisible_to_the_caller(monkeypatch, precision):
    """fp32 / mixed contexts physically re-order the bodies into sorted order every 16th build
    (BH_REORDER_EVERY); everything host-facing must keep speaking the caller's order: positions,
    velocities, masses, accelerations, the occupant indices of the exported tree."""
    from gpu_nbody_simulation_amd import initial_conditions as IC
    n = 30011
    m, p, v = IC.make("plummer", n, 4)
    m = m * np.linspace(1e-4, 2e-4, n)                        # distinct masses: a permutation would show
    res = {}
    for every in ("0", "16", "1"):
        monkeypatch.setenv("BH_REORDER_EVERY", every)
        with G.BarnesHutEngine(G.BhConfig(capacity=n, precision=precision, max_depth=18, reference_compat=False)) as e:
            e.upload(p, v, m)
            e.step(35)                                        # re-ordered at builds 0, 16, 32 (every = 16)
            pos, vel = e.download()
            e.compute_forces()
            acc = e.accelerations()
            frc = e.forces()
            mass = e.masses()
            nodes, depth = e.export_tree()
        res[every] = (pos, vel, acc, mass, nodes)
        expect_m = m if precision == G.Precision.MIXED else m.astype(np.float32).astype(np.float64)
        assert np.array_equal(mass, expect_m)
        np.testing.assert_allclose(frc, acc * mass[:, None], rtol=1e-12)
        # a single-occupant leaf names a body whose (caller-order) position lies inside the leaf's box
        leaf = nodes[nodes["particle"] >= 0]
        idx = leaf["particle"].astype(np.int64)
        assert len(np.unique(idx)) == len(idx) > n // 2
        q = pos[idx]
        assert ((q[:, 0] >= leaf["xmin"]) & (q[:, 0] <= leaf["xmax"]) & (q[:, 1] >= leaf["ymin"]) & (q[:, 1] <= leaf["ymax"])).all()
    for every in ("16", "1"):
        for a, b in zip(res["0"][:3], res[every][:3]):
            # same bodies, same forces; only the order of a few fp32 sums (equal-key ties) may differ
            np.testing.assert_allclose(b, a, rtol=2e-4, atol=1e-12)
        assert np.median(np.abs(res[every][0] - res["0"][0])) == 0.0
