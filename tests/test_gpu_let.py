"""Distributed step with locally-essential trees, rehearsed on ONE GPU: W contexts act as W ranks,
the two collectives (all_gather of bounds, all_to_all of LET blocks) are emulated by device copies.

What must hold:
  * theta -> 0: every node is openable, the LETs are whole trees and the forest walk is the direct
    sum over all bodies -> equals the oracle's direct sum to fp32 rounding;
  * theta = 0.5: the forest walk is a Barnes-Hut evaluation of its own (cells that straddle two
    ranks are split into per-rank partial cells), so it is compared with the direct sum and must be
    as accurate as the single-tree walk; against the single-tree walk it differs by ~1e-3;
  * LETs are much smaller than the trees, nothing overflows, several steps stay finite and follow
    the single-context trajectory."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import bh_oracle as O  # noqa: E402
import gpu_nbody_simulation_amd as G  # noqa: E402
from gpu_nbody_simulation_amd import initial_conditions as IC  # noqa: E402
from gpu_nbody_simulation_amd.distributed import partition_hilbert, partition_orb, wrap_device  # noqa: E402


class EmulatedRanks:
    def __init__(self, mass, pos, vel, world, let_cap, partition=partition_orb, **cfg):
        self.world = world
        self.parts = partition(pos, world)
        dev = torch.device("cuda", 0)
        self.engs, self.bufs = [], []
        cfg.setdefault("precision", G.Precision.F32)
        # every context sized for ITS OWN bodies, as bench.py does: the contexts' quad arrays then differ
        # in size, and only the agreed forest_base makes a sender's links land in the receiver's blocks
        for ix in self.parts:
            e = G.BarnesHutEngine(G.BhConfig(capacity=max(len(ix), 1), **cfg))
            e.upload(pos[ix], vel[ix], mass[ix])
            self.engs.append(e)
        self.forest_base = max(e.let_local_quads() for e in self.engs)
        for r, e in enumerate(self.engs):
            e.let_configure(r, world, let_cap, self.forest_base)
            lb, ab, sd, rv, nb, k = e.let_pointers()
            self.bufs.append((wrap_device(lb, 4 * k, "<f8", dev), wrap_device(ab, 4 * k * world, "<f8", dev),
                              wrap_device(sd, world * nb, "|u1", dev), wrap_device(rv, world * nb, "|u1", dev), nb))

    def step(self, integrate=True, two_launches=False):
        for e in self.engs:
            e.let_bounds()
            e.sync()
        allb = torch.cat([b[0] for b in self.bufs])                 # "all_gather"
        for b in self.bufs:
            b[1].copy_(allb)
        torch.cuda.synchronize()
        for e in self.engs:
            e.let_build()
            e.sync()
        for r in range(self.world):                                  # "all_to_all"
            nb = self.bufs[r][4]
            for q in range(self.world):
                if q != r:
                    self.bufs[q][3][r * nb:(r + 1) * nb].copy_(self.bufs[r][2][q * nb:(q + 1) * nb])
        torch.cuda.synchronize()
        for e in self.engs:
            if two_launches:
                e.let_walk_local()
                e.let_walk_remote(integrate)
            else:
                e.let_walk() if integrate else e.let_forces()
            e.sync()

    def gather(self, what):
        n = sum(len(ix) for ix in self.parts)
        out = np.zeros((n, 2))
        for e, ix in zip(self.engs, self.parts):
            out[ix] = what(e)
        return out

    def close(self):
        for e in self.engs:
            e.close()


def rel(a, ref):
    return np.linalg.norm(a - ref, axis=1) / np.linalg.norm(ref, axis=1)


@pytest.mark.parametrize("world", [2, 5])
def test_theta_zero_forest_is_the_direct_sum(world):
    n = 3000
    m, p, v = IC.make("uniform", n, 9)
    ref = O.direct_forces(p, m) / m[:, None]
    er = EmulatedRanks(m, p, v, world, let_cap=8192, max_depth=21, theta=1e-6, reference_compat=False)
    er.step(integrate=False)
    a = er.gather(lambda e: e.accelerations())
    counts = [e.let_counts() for e in er.engs]
    er.close()
    r = rel(a, ref)
    assert np.median(r) < 5e-6 and r.max() < 1e-3
    # theta -> 0: the LET sent to a peer is the sender's whole tree
    for rk, c in enumerate(counts):
        assert c[rk] == 0 and all(x > 0 for i, x in enumerate(c) if i != rk)


@pytest.mark.parametrize("kind,world", [("plummer", 4), ("uniform", 8)])
def test_forest_walk_accuracy_and_let_size(kind, world):
    n = 65536
    m, p, v = IC.make(kind, n, 3)
    with G.BarnesHutEngine(G.BhConfig(capacity=n, precision=G.Precision.F32, max_depth=21, reference_compat=False)) as e:
        e.upload(p, v, m)
        e.compute_forces()
        a_single = e.accelerations()
        n_quads_single = e.stats().n_internal + 1
    er = EmulatedRanks(m, p, v, world, let_cap=16384, max_depth=21, reference_compat=False)
    er.step(integrate=False)
    a = er.gather(lambda e: e.accelerations())
    counts = np.array([e.let_counts() for e in er.engs])
    er.close()
    assert np.isfinite(a).all()
    # ground truth on a sample: direct sum (oracle, fp64)
    idx = np.arange(0, n, 97)
    sub = np.zeros((len(idx), 2))
    for k, i in enumerate(idx):
        d = p - p[i]
        r2 = (d ** 2).sum(1)
        r2[i] = np.inf
        sub[k] = (6.67e-11 * m[:, None] * d / (r2 ** 1.5)[:, None]).sum(0)
    e_forest = rel(a[idx], sub)
    e_single = rel(a_single[idx], sub)
    # the forest walk is at least as accurate as the single-tree walk (partial cells are finer)
    assert np.median(e_forest) <= 1.2 * np.median(e_single) + 1e-6
    assert np.quantile(e_forest, 0.99) <= 1.5 * np.quantile(e_single, 0.99) + 1e-5
    # and close to it
    assert np.median(rel(a, a_single)) < 3e-3
    # LETs are a small part of the trees
    per_pair = counts[counts > 0]
    assert per_pair.max() < 16384 and per_pair.mean() < 0.5 * n_quads_single / world * 2
    print(kind, world, "LET quads per pair: mean %.0f max %d; local tree quads ~%d" % (per_pair.mean(), per_pair.max(), n_quads_single // world))


def test_steps_follow_the_single_context_run():
    n = 20000
    m, p, v = IC.make("uniform", n, 11, quasi_static=False)
    m = m * 1e-3                                   # gentle dynamics: no close-encounter blow-ups in 5 steps
    cfg = dict(max_depth=21, reference_compat=False)
    with G.BarnesHutEngine(G.BhConfig(capacity=n, precision=G.Precision.F32, **cfg)) as e:
        e.upload(p, v, m)
        e.step(5)
        ps, vs = e.download()
    er = EmulatedRanks(m, p, v, 3, let_cap=16384, **cfg)
    for _ in range(5):
        er.step()
    pf = er.gather(lambda e: e.download()[0])
    vf = er.gather(lambda e: e.download()[1])
    er.close()
    assert np.isfinite(pf).all()
    dv_s, dv_f = vs - v, vf - v
    r = np.linalg.norm(dv_f - dv_s, axis=1) / np.linalg.norm(dv_s, axis=1)
    assert np.median(r) < 5e-3
    # fp32 positions (ulp ~7e-9 here); a handful of bodies in close pairs amplify the ~1e-3 force difference
    d = np.abs(pf - ps).max(axis=1)
    assert np.quantile(d, 0.999) < 1e-6 and d.max() < 1e-4


def test_let_overflow_is_reported():
    n = 20000
    m, p, v = IC.make("uniform", n, 2)
    er = EmulatedRanks(m, p, v, 2, let_cap=16, max_depth=21, reference_compat=False)
    er.step(integrate=False)
    with pytest.raises(G.BhError) as ei:
        er.engs[0].let_counts()
    assert ei.value.code == -4
    er.close()


def _two_clouds(rng, n, gap):
    """Two Gaussian clouds of n bodies (sigma 0.01) whose centres are `gap` apart: far apart the peer opens
    only the top of the other tree (a small LET), interpenetrating it opens most of it (a large one)."""
    a = rng.normal(0.0, 0.01, (n, 2)) + [-gap / 2, 0.0]
    b = rng.normal(0.0, 0.01, (n, 2)) + [gap / 2, 0.0]
    return [x.astype(np.float32).astype(np.float64) for x in (a, b)]


def test_let_overflow_of_a_middle_step_is_still_reported():
    """the overflow flag and the counts used to be cleared by every build, so a check every N
    steps saw the last build only.  Three builds -- small LETs, LETs far beyond let_cap, small again --
    and ONE read afterwards: it must report the overflow and the largest count; the next interval is clean."""
    rng = np.random.default_rng(3)
    n = 6000
    m = np.full(2 * n, 1e-3)
    v = np.zeros((2 * n, 2))
    far, near = _two_clouds(rng, n, 2.0), _two_clouds(rng, n, 0.004)
    split = lambda pp, w: [np.arange(0, n), np.arange(n, 2 * n)]

    def run(er, states):
        for st in states:
            for e, pos in zip(er.engs, st):
                e.upload(pos, v[:n], m[:n])
            er.step(integrate=False)

    probe = EmulatedRanks(m, np.concatenate(far), v, 2, let_cap=1 << 15, partition=split, max_depth=21, reference_compat=False)
    run(probe, [far])
    small = max(max(e.let_counts()) for e in probe.engs)
    run(probe, [near])
    large = max(max(e.let_counts()) for e in probe.engs)
    probe.close()
    assert large > 8 * small, (small, large)
    cap = 2 * small
    er = EmulatedRanks(m, np.concatenate(far), v, 2, let_cap=cap, partition=split, max_depth=21, reference_compat=False)
    run(er, [far, near, far])                                  # nothing read in between
    for e in er.engs:
        counts, ov = e.let_counts(with_overflow=True)
        assert ov and max(counts) == pytest.approx(large, rel=0.02) and max(counts) > cap
    run(er, [far])
    for e in er.engs:
        counts, ov = e.let_counts(with_overflow=True)          # reading started a new interval
        assert not ov and max(counts) <= small
    er.close()


def test_local_tree_overflow_is_reported_through_the_let_counters():
    """when the local tree outgrows node_capacity, let_mark / let_pack leave the send blocks as
    they were and the peers would walk stale data; the rank's own LET counters must say so (they are what
    LetStepper.check() all-reduces)."""
    rng = np.random.default_rng(4)
    n = 4000
    pos = _two_clouds(rng, n, 0.5)
    m, v = np.full(n, 1e-3), np.zeros((n, 2))
    dev = torch.device("cuda", 0)
    engs = [G.BarnesHutEngine(G.BhConfig(capacity=n, precision=G.Precision.F32, max_depth=21, reference_compat=False,
                                         node_capacity=(4001 if r == 0 else 0))) for r in range(2)]
    fb = max(e.let_local_quads() for e in engs)
    bufs = []
    for r, e in enumerate(engs):
        e.upload(pos[r], v, m)
        e.let_configure(r, 2, 4096, fb)
        lb, ab, sd, rv, nb, k = e.let_pointers()
        bufs.append((wrap_device(lb, 4 * k, "<f8", dev), wrap_device(ab, 4 * k * 2, "<f8", dev)))
    for e in engs:
        e.let_bounds()
    torch.cuda.synchronize()
    allb = torch.cat([b[0] for b in bufs])
    for b in bufs:
        b[1].copy_(allb)
    torch.cuda.synchronize()
    for e in engs:
        e.let_build()
    torch.cuda.synchronize()
    assert engs[0].let_counts(with_overflow=True)[1]             # rank 0: 1,000 quads cannot hold 4,000 bodies
    assert not engs[1].let_counts(with_overflow=True)[1]
    with pytest.raises(G.BhError):
        engs[0].sync()                                           # and bh_sync names the cause
    for e in engs:
        e.close()


def test_ranks_without_bodies():
    n = 3
    m, p, v = IC.make("uniform", n, 4)
    ref = O.direct_forces(p, m) / m[:, None]
    er = EmulatedRanks(m, p, v, 5, let_cap=64, max_depth=21, reference_compat=False)
    assert min(len(ix) for ix in er.parts) == 0
    er.step(integrate=False)
    a = er.gather(lambda e: e.accelerations())
    er.close()
    assert rel(a, ref).max() < 1e-5


@pytest.mark.parametrize("no_split", [True, False])
def test_two_launch_forest_walk_equals_the_single_launch(no_split):
    """bh_let_walk_local + bh_let_walk_remote (what lets the all_to_all overlap the local walk) against
    bh_let_walk: the same sums in the same order with one wavefront per group (bitwise), summation order
    only with the level-synchronous walk."""
    from gpu_nbody_simulation_amd.engine import FLAG_WALK_NO_SPLIT
    n = 50000
    m, p, v = IC.make("plummer", n, 12)
    m = m * 1e-3
    out = []
    for two in (False, True):
        er = EmulatedRanks(m, p, v, 3, let_cap=16384, max_depth=21, reference_compat=False,
                           flags=FLAG_WALK_NO_SPLIT if no_split else 0)
        er.step(integrate=False, two_launches=two)
        a = er.gather(lambda e: e.accelerations())
        er.step(two_launches=two)
        er.step(two_launches=two)
        out.append((a, er.gather(lambda e: e.download()[0]), er.gather(lambda e: e.download()[1])))
        er.close()
    (a1, p1, v1), (a2, p2, v2) = out
    if no_split:
        assert np.array_equal(a1, a2) and np.array_equal(p1, p2) and np.array_equal(v1, v2)
    else:
        r = rel(a2, a1)
        assert np.median(r) < 5e-7 and np.quantile(r, 0.999) < 2e-5
        assert np.abs(p2 - p1).max() < 1e-6


def test_contexts_of_different_capacity_need_the_agreed_forest_base():
    """The layout bug this guards against: a sender used its OWN local-quad count as the start of the
    receiver's blocks, which is only right when all contexts have the same capacity."""
    n = 20000
    m, p, v = IC.make("plummer", n, 3)
    er = EmulatedRanks(m, p, v, 3, let_cap=16384, partition=lambda pp, w: [np.arange(0, 3000), np.arange(3000, 12000), np.arange(12000, n)],
                       max_depth=21, reference_compat=False)
    assert len({e.let_local_quads() for e in er.engs}) == 3          # three different array sizes
    er.step(integrate=False)
    a = er.gather(lambda e: e.accelerations())
    for e in er.engs:
        e.let_counts()                                                # (raises if a LET did not fit)
    with pytest.raises(G.BhError):                                    # a smaller base than the largest context's is refused
        small = min(e.let_local_quads() for e in er.engs)
        big = max(er.engs, key=lambda e: e.let_local_quads())
        big.let_configure(er.engs.index(big), 3, 16384, small)
    er.close()
    ref = O.direct_forces(p, m) / m[:, None]
    with G.BarnesHutEngine(G.BhConfig(capacity=n, precision=G.Precision.F32, max_depth=21, reference_compat=False)) as e:
        e.upload(p, v, m)
        e.compute_forces()
        r1 = rel(e.accelerations(), ref)
    r = rel(a, ref)
    # the Barnes-Hut error of a single tree at theta 0.5 (monopoles), not garbage
    assert np.median(r) <= 1.2 * np.median(r1) and np.quantile(r, 0.99) <= 1.5 * np.quantile(r1, 0.99)
