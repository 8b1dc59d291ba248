"""SURVEY 8(f) rows on the GPU: on-device initialisers, trajectory writer, scaling-script shim."""
import os
import re
import shutil
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import bh_oracle as O  # noqa: E402
import gpu_nbody_simulation_amd as G  # noqa: E402
from gpu_nbody_simulation_amd.engine import FLAG_WALK_NO_SPLIT  # noqa: E402
from gpu_nbody_simulation_amd import project, scaling, textio  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.mark.parametrize("precision", [G.Precision.F64_EXACT, G.Precision.F32])
def test_device_initialiser_box_distribution(precision):
    """initializeGpu (project.cu:304-341) with the reference's ranges (project.cu:30-35): masses
    log-uniform on [0.1, 0.5] (both bounds positive, project.cu:86-89), vectors linear."""
    n = 200000
    with G.BarnesHutEngine(G.BhConfig(capacity=n, precision=precision)) as e:
        e.initialize(n, seed=7)
        p, v = e.download()
        m = e.masses()
        e.initialize(n, seed=7)
        p2, v2 = e.download()
        e.initialize(n, seed=8)
        p3, _ = e.download()
    assert np.array_equal(p, p2) and np.array_equal(v, v2) and not np.array_equal(p, p3)
    assert 0.1 <= m.min() and m.max() <= 0.5 and -0.1 <= p.min() and p.max() <= 0.1 and np.abs(v).max() <= 1e-4
    lm = np.log10(m)
    assert abs(lm.mean() - (np.log10(0.1) + np.log10(0.5)) / 2) < 2e-3          # log-uniform
    assert abs(p.mean()) < 1e-3 and abs(p.std() - 0.2 / np.sqrt(12)) < 5e-4      # uniform
    assert abs(np.corrcoef(p[:, 0], p[:, 1])[0, 1]) < 0.01                       # independent streams
    assert len(np.unique(p[:, 0])) > 0.99 * n or precision == G.Precision.F32


def test_device_initialiser_plummer():
    n = 200000
    with G.BarnesHutEngine(G.BhConfig(capacity=n, precision=G.Precision.F32, max_depth=21, reference_compat=False)) as e:
        e.initialize(n, seed=1, kind="plummer", higher_m=1e-8 / n, lower_p=0.02, higher_p=0.2)
        p, v = e.download()
        m = e.masses()
        e.step(2)
        p2, _ = e.download()
    r = np.hypot(p[:, 0], p[:, 1])
    assert r.max() <= 0.2 * (1 + 1e-6) and not v.any() and np.allclose(m, 1e-8 / n, rtol=1e-6)
    assert 0.49 < (r < 0.02).mean() < 0.53
    assert np.isfinite(p2).all()


def test_initializeGpu_and_saved_files_round_trip(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    m, p, v = project.initializeGpu(1000, seed=3, save_to_file=True)
    m2, p2, v2 = textio.loadSimulationDataFromText("masses_init.txt", "positions_init.txt",
                                                   "velocities_init.txt", 1000, verbose=False)
    assert np.allclose(m2, m, rtol=1e-5) and np.allclose(p2, p, rtol=1e-5, atol=1e-12)   # 6 significant digits


def test_trajectory_file_matches_the_oracle(tmp_path, init1024):
    """positions file as runSimulationCpu writes it (savePositions, project.cu:855-863, 876, 909)."""
    m, p, v = init1024
    project.runSimulationGpu(m, p, v, 3, out_dir=str(tmp_path), positions_file="positions_gpu.txt")
    lines = (tmp_path / "positions_gpu.txt").read_text().splitlines()
    assert len(lines) == 4 * 1024
    want = []
    pp, vv = p.copy(), v.copy()
    for s in range(4):
        want += ["%f %d %f %f " % (float(s), i, x, y) for i, (x, y) in enumerate(pp)]
        pp, vv = O.run(pp, vv, m, 1, max_depth=10)
    assert lines == want


def test_reference_scaling_script_lines_run_unchanged(tmp_path):
    """The two lines of first_scaling_script.sh:30,33 with the stand-in first on PATH."""
    for f in ("masses", "positions", "velocities"):
        shutil.copy(os.path.join(GOLD, "init1024", f"{f}_init.txt"), tmp_path / f"{f}_init.txt")
    env = dict(os.environ, PATH=os.path.join(ROOT, "gpu-nbody-simulation_amd", "compat") + os.pathsep + os.environ["PATH"])
    script = ('set -e\n'
              'nvcc -DN_BODIES=1024 -DN_THREADS=64 -DN_SIMULATIONS=5 -o project project.cu\n'
              'runtime=$(./project)\n'
              'echo "1024, 64, 5, $runtime" >> results.txt\n')
    subprocess.check_call(["bash", "-c", script], cwd=tmp_path, env=env)
    recs = scaling.parse_results(str(tmp_path / "results.txt"))
    assert len(recs) == 1 and recs[0]["n_bodies"] == 1024 and recs[0]["parallel_us"] > 0
    assert os.path.getsize(tmp_path / "quadtree_init_gpu.txt") > 0


@pytest.mark.parametrize("precision", [G.Precision.F64_EXACT, G.Precision.F32, G.Precision.F64])
def test_n_threads_limits_the_bodies_walked_at_a_time_and_changes_no_result(precision, init1024):
    """N_THREADS (project.cu:5-7, 703): the walk takes the bodies in passes of n_threads (whole 256-thread workgroups),
    one launch after the other -- the axis of the reference's first scaling experiment (first_scaling_script.sh:17-36).
    Same bodies, same tree, same per-body sums: results bitwise equal for every n_threads; the number of walk launches per
    step is what changes (bh_stats_t.walk_launches)."""
    m, p, v = init1024
    if precision == G.Precision.F32:
        m, p, v = (x.astype(np.float32).astype(np.float64) for x in (m, p, v))
    out, launches = [], []
    for nt in (0, 1, 300, 1024):
        # (fp32: one wavefront per 64 bodies on both sides -- a pass is one thread per body; without the cap a launch
        # this small would let 8 wavefronts share each group, another order of the fp32 sums)
        with G.BarnesHutEngine(G.BhConfig(capacity=1024, precision=precision, n_threads=nt, flags=FLAG_WALK_NO_SPLIT,
                                          max_depth=10 if precision != G.Precision.F32 else 16)) as e:
            e.upload(p, v, m)
            e.step(3)
            out.append(e.download())
            launches.append(e.stats().walk_launches)
    for pp, vv in out[1:]:
        assert np.array_equal(pp, out[0][0]) and np.array_equal(vv, out[0][1])
    # the mechanism, not a timing: passes of ceil(n_threads / 256) workgroups, one launch each
    assert launches == [1, 4, 2, 1]


def test_overflow_is_reported_by_sync_after_step():
    r = np.random.default_rng(0)
    n = 4096
    m, p, v = 10.0 ** r.uniform(-2, 1, n), r.uniform(-0.1, 0.1, (n, 2)), r.uniform(-1e-4, 1e-4, (n, 2))
    with G.BarnesHutEngine(G.BhConfig(capacity=n, node_capacity=101, precision=G.Precision.F32, max_depth=16)) as e:
        e.upload(p, v, m)
        e.step(1)
        with pytest.raises(G.BhError) as ei:
            e.sync()
        assert ei.value.code == -4


@pytest.mark.parametrize("precision", [G.Precision.F32, G.Precision.F64, G.Precision.F64_EXACT])
@pytest.mark.parametrize("nsteps", [2, 3])
def test_overflow_inside_a_multi_step_call_keeps_the_last_good_state(precision, nsteps):
    """a tree that outgrows node_capacity in step s of bh_step(k) makes that step's walk return at once -- and it
    used to leave the bounds slots at +-inf, from which step s + 1 built a box of {+inf, -inf}, a short garbage tree that
    did NOT overflow, and integrated every body with wrong forces; bh_sync then said BH_OK.  Now the empty reduction keeps
    the box and the flag: every step of the call is a no-op, sync and download raise -4, and the device state is the
    uploaded one, bit for bit."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")

    def peek(ptr, count, dtype):                                               # (bh_download refuses, rightly: read the device arrays directly)
        out = np.empty(count, dtype=dtype)
        assert hip.hipMemcpy(C.c_void_p(out.ctypes.data), C.c_void_p(ptr), C.c_size_t(out.nbytes), 2) == 0    # hipMemcpyDeviceToHost
        return out.astype(np.float64)
    r = np.random.default_rng(0)
    n = 4096
    m, p, v = 10.0 ** r.uniform(-2, 1, n), r.uniform(-0.1, 0.1, (n, 2)), r.uniform(-1e-4, 1e-4, (n, 2))
    f32 = precision == G.Precision.F32
    if f32:
        m, p, v = (x.astype(np.float32).astype(np.float64) for x in (m, p, v))
    with G.BarnesHutEngine(G.BhConfig(capacity=n, node_capacity=101, precision=precision, max_depth=16)) as e:
        e.upload(p, v, m)
        e.step(nsteps)
        with pytest.raises(G.BhError) as ei:
            e.sync()
        assert ei.value.code == -4
        with pytest.raises(G.BhError) as ei:
            e.download()
        assert ei.value.code == -4
        dp, dv, _, dn, eb = e.device_state()
        assert dn == n and eb == (4 if f32 else 8)
        ts = np.float32 if f32 else np.float64
        gp, gv = peek(dp, 2 * n, ts).reshape(n, 2), peek(dv, 2 * n, ts).reshape(n, 2)
    # (fp32 mode keeps its state in the sorted order of its first build: compare as sets of (x, y, vx, vy) rows)
    a = np.hstack([gp, gv]); b = np.hstack([p, v])
    a = a[np.lexsort(a.T[::-1])]; b = b[np.lexsort(b.T[::-1])]
    assert np.array_equal(a, b)
