"""World-size-2 run of the product's sharding/exchange logic (gpu_nbody_simulation_amd.distributed)
over gloo on the CPU.  Compute comes from a stand-in engine built on the oracle (tests/ only); what
is under test is ownership, the in-place all_gather of fixed-size blocks and the scatter back."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs(n, seed=7):
    rng = np.random.default_rng(seed)
    return ((10.0 ** rng.uniform(-2, 1, n)).astype(np.float32), rng.uniform(-0.1, 0.1, (n, 2)).astype(np.float32),
            rng.uniform(-1e-4, 1e-4, (n, 2)).astype(np.float32))


def _worker(rank, world, port, n, steps, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_standin import OracleStandInEngine
    from gpu_nbody_simulation_amd.distributed import ShardedStepper
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, p, v = _inputs(n)
    eng = OracleStandInEngine()
    eng.upload(p, v, m)
    st = ShardedStepper(eng, rank, world, n, torch.device("cpu"))
    lo, hi = st.lo, st.hi
    for _ in range(steps):
        st.step()
    pos, vel = eng.download()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), pos=pos, vel=vel, lo=lo, hi=hi)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [512, 700])         # chunks of 256: full/full and full/partial
def test_two_ranks_equal_one_rank(tmp_path, n):
    world, steps = 2, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, steps, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    # ownership: disjoint, contiguous, covering, fixed-size chunks of ceil(n/world)
    chunk = ((n + world - 1) // world + 255) // 256 * 256
    assert (int(r0["lo"]), int(r0["hi"])) == (0, chunk) and (int(r1["lo"]), int(r1["hi"])) == (chunk, n)
    # replicas agree after the exchange
    assert np.array_equal(r0["pos"], r1["pos"]) and np.array_equal(r0["vel"], r1["vel"])
    # and equal the single-process run bit for bit (a body's walk does not depend on who runs it)
    from dist_standin import OracleStandInEngine
    m, p, v = _inputs(n)
    ref = OracleStandInEngine()
    ref.upload(p, v, m)
    ref.set_owned_fraction(0, 1)
    ref.step(steps)
    pos, vel = ref.download()
    assert np.array_equal(pos, r0["pos"]) and np.array_equal(vel, r0["vel"])
    assert not np.array_equal(pos, p.astype(np.float64))


def test_world_one_uses_plain_step():
    from dist_standin import OracleStandInEngine
    from gpu_nbody_simulation_amd.distributed import ShardedStepper
    m, p, v = _inputs(32)
    eng = OracleStandInEngine()
    eng.upload(p, v, m)
    st = ShardedStepper(eng, 0, 1, 32, torch.device("cpu"))
    assert (st.lo, st.hi) == (0, 32)
    st.step()
    assert not np.array_equal(eng.download()[0], p.astype(np.float64))


# ---- LetStepper: all_gather of bounds + all_to_all of fixed-size LET blocks --------------------------
def _let_worker(rank, world, port, n, steps, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_standin import LetStandInEngine
    from gpu_nbody_simulation_amd.distributed import LetStepper, partition_hilbert
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, p, v = _inputs(n)
    from gpu_nbody_simulation_amd.distributed import partition_orb
    mine = partition_orb(p, world)[rank]
    eng = LetStandInEngine()
    eng.upload(p[mine], v[mine], m[mine])
    st = LetStepper(eng, rank, world, let_cap=2, device=torch.device("cpu"), ids=mine)   # far too small on purpose
    st.step(integrate=False)
    overflowed = False
    try:
        st.check()
    except RuntimeError:
        overflowed = True
    cap = st.autotune()
    for k in range(steps):
        st.step()
        if k == 0:
            # re-deal the bodies after the first step (ORB cuts from the distributed histogram, bodies moved
            # by all_to_all): the trajectory must not notice who owns what
            held = st.rebalance()
            cap = st.let_cap
            assert held == eng.n and st.cuts is not None
            own = st.cuts.owner(eng.download()[0])
            assert (own == rank).all()                     # every body sits on the rank the cut tree names
    largest = st.check()
    st.run(0)
    pos, vel = eng.download()
    mine = st.ids
    np.savez(os.path.join(out_dir, f"let{rank}.npz"), pos=pos, vel=vel, idx=mine, cap=cap, largest=largest,
             forest_base=eng.forest_base_seen, local_quads=eng.let_local_quads(),
             overflowed=overflowed, bounds=eng.seen_bounds)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 300), (3, 200)])
def test_let_stepper_ranks_reproduce_the_direct_sum(tmp_path, world, n):
    steps = 2
    port = _free_port()
    mp.spawn(_let_worker, args=(world, port, n, steps, str(tmp_path)), nprocs=world, join=True)
    from oracle import bh_oracle as O
    m, p, v = _inputs(n)
    p64, v64, m64 = p.astype(np.float64), v.astype(np.float64), m.astype(np.float64)
    pos, vel = p.copy(), v.copy()
    for _ in range(steps):                                 # single-process fp32 state, fp64 direct sum
        a = O.direct_forces(pos.astype(np.float64), m64) / m64[:, None]
        vel = vel + a.astype(np.float32)
        pos = pos + vel
    got_p, got_v = np.zeros((n, 2)), np.zeros((n, 2))
    seen = np.zeros(n, dtype=int)
    caps = set()
    for r in range(world):
        d = np.load(tmp_path / f"let{r}.npz")
        got_p[d["idx"]], got_v[d["idx"]] = d["pos"], d["vel"]
        seen[d["idx"]] += 1
        caps.add(int(d["cap"]))
        assert bool(d["overflowed"])                       # let_cap=2 was reported as too small ...
        assert int(d["largest"]) <= int(d["cap"])          # ... and autotune fixed it
        # every rank saw every rank's bounds, in rank order
        b = d["bounds"]
        assert b.shape == (world, 8, 4) and np.isfinite(b).all() and (b[..., 0] <= b[..., 1]).all()
        if r:
            assert np.array_equal(b, first_bounds)
        first_bounds = b
    # all ranks configured the same forest_base: the largest local-quad count of any rank
    fbs = [int(np.load(tmp_path / f"let{r}.npz")["forest_base"]) for r in range(world)]
    assert len(set(fbs)) == 1 and fbs[0] >= max(int(np.load(tmp_path / f"let{r}.npz")["local_quads"]) for r in range(world))
    assert (seen == 1).all()                               # the partition covers every body once
    assert len(caps) == 1                                  # all ranks agreed on the new block size
    np.testing.assert_allclose(got_v, vel.astype(np.float64), rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(got_p, pos.astype(np.float64), rtol=2e-6, atol=1e-9)
    assert not np.array_equal(got_p, p64)


def _rebalance_worker(rank, world, port, n, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_standin import LetStandInEngine
    from gpu_nbody_simulation_amd.distributed import LetStepper
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, p, v = _inputs(n)
    mine = np.arange(rank, n, world)                       # every rank generated "its share": no spatial meaning at all
    eng = LetStandInEngine()
    eng.upload(p[mine], v[mine], m[mine])
    st = LetStepper(eng, rank, world, let_cap=4096, device=torch.device("cpu"), ids=mine)
    held = [st.rebalance()]
    st.step()
    held.append(st.rebalance())                             # a second time on the moved bodies
    pos, vel = eng.download()
    np.savez(os.path.join(out_dir, f"reb{rank}.npz"), pos=pos, vel=vel, idx=st.ids, mass=eng.masses(), held=held,
             axis=st.cuts.axis, value=st.cuts.value, box=st.cuts.box)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_rebalance_deals_arbitrary_shares_into_balanced_orb_domains(tmp_path, world):
    """no rank ever holds all bodies.  Every rank starts with an arbitrary share (every
    world-th body); rebalance() derives the ORB cuts from the all-reduced histograms and moves the bodies by
    all_to_all: afterwards every body is on the rank its position selects, nobody is lost or duplicated,
    ids, masses and velocities travelled with their bodies, all ranks hold the same cut tree, the shares are
    balanced, and the trajectory is the single-process one."""
    n = 600
    port = _free_port()
    mp.spawn(_rebalance_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    from oracle import bh_oracle as O
    from gpu_nbody_simulation_amd.distributed import OrbCuts
    m, p, v = _inputs(n)
    m64 = m.astype(np.float64)
    a = O.direct_forces(p.astype(np.float64), m64) / m64[:, None]
    vel = v + a.astype(np.float32)
    pos = p + vel
    d = [np.load(tmp_path / f"reb{r}.npz") for r in range(world)]
    idx = np.concatenate([x["idx"] for x in d])
    assert np.array_equal(np.sort(idx), np.arange(n))       # conserved: every id exactly once
    got_p = np.concatenate([x["pos"] for x in d])
    got_v = np.concatenate([x["vel"] for x in d])
    got_m = np.concatenate([x["mass"] for x in d])
    np.testing.assert_array_equal(got_m, m64[idx])          # masses followed their ids
    np.testing.assert_allclose(got_v, vel.astype(np.float64)[idx], rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(got_p, pos.astype(np.float64)[idx], rtol=2e-6, atol=1e-9)
    for r in range(1, world):                               # one cut tree everywhere
        assert np.array_equal(d[r]["axis"], d[0]["axis"]) and np.array_equal(d[r]["value"], d[0]["value"])
    cuts = OrbCuts(world, d[0]["box"])
    cuts.axis[:], cuts.value[:] = d[0]["axis"], d[0]["value"]
    for r in range(world):
        assert (cuts.owner(d[r]["pos"]) == r).all()         # ownership = the cut tree
        assert abs(len(d[r]["idx"]) - n / world) <= 0.08 * n / world + 2, [len(x["idx"]) for x in d]
        assert list(d[r]["held"]) [-1] == len(d[r]["idx"])


def test_orb_cut_tree_numpy_twin():
    """OrbCuts.descend / histogram / choose_cut: pre-order cut layout for any world size, histograms add up,
    the chosen cut is a tree-grid line that balances the two sides."""
    from gpu_nbody_simulation_amd.distributed import ORB_BINS, OrbCuts, choose_cut, padded_root_box
    rng = np.random.default_rng(1)
    p = np.concatenate([rng.normal(0, 0.02, (15000, 2)), rng.uniform(-0.2, 0.2, (5000, 2))])
    w = rng.integers(1, 50, len(p))
    for world in (1, 2, 3, 6, 8):
        c = OrbCuts(world, padded_root_box(p[:, 0].min(), p[:, 0].max(), p[:, 1].min(), p[:, 1].max()))
        for level in range(c.depth()):
            regs = c.regions(level)
            for k, _, _, rb in regs:
                c.axis[k] = int((rb[3] - rb[2]) > (rb[1] - rb[0]))
            h = c.histogram(p, w, level)
            r0, kk, nr = c.descend(p, level)
            assert h.sum() == w[nr > 1].sum()
            for k, r_first, nr_k, rb in regs:
                val = choose_cut(h[k], rb, c.box, int(c.axis[k]), (nr_k // 2) / nr_k)
                c.value[k] = val
                ax = int(c.axis[k])
                e = (val - c.box[2 * ax]) / (c.box[2 * ax + 1] - c.box[2 * ax]) * ORB_BINS
                assert abs(e - round(e)) < 1e-6 and rb[2 * ax] < val < rb[2 * ax + 1]   # a grid line inside the region
        own = c.owner(p)
        assert own.min() >= 0 and own.max() == world - 1
        share = np.array([w[own == r].sum() for r in range(world)]) / w.sum()
        assert np.abs(share - 1.0 / world).max() <= 0.06 / world + 0.01, share          # weighted balance


def test_partition_orb_is_a_balanced_partition_into_disjoint_boxes():
    from gpu_nbody_simulation_amd.distributed import partition_orb
    rng = np.random.default_rng(0)
    p = rng.normal(size=(1000, 2))
    for world in (1, 2, 3, 5, 8):
        parts = partition_orb(p, world)
        allidx = np.sort(np.concatenate(parts))
        assert np.array_equal(allidx, np.arange(1000))
        sizes = [len(x) for x in parts]
        assert max(sizes) - min(sizes) <= 0.05 * 1000 / world + world   # a snapped cut moves <= 1 % of a piece's bodies
        exact = [len(x) for x in partition_orb(p, world, snap=False)]
        assert max(exact) - min(exact) <= world
        boxes = [(p[ix].min(0), p[ix].max(0)) for ix in parts]
        for i in range(world):
            for j in range(i + 1, world):
                (alo, ahi), (blo, bhi) = boxes[i], boxes[j]
                overlap = np.all(np.minimum(ahi, bhi) - np.maximum(alo, blo) > 0)
                assert not overlap


def test_partition_hilbert_is_a_partition_into_compact_curve_ranges():
    from gpu_nbody_simulation_amd.distributed import hilbert_index, partition_hilbert
    # the index is a bijection of the grid and consecutive cells are neighbours
    g = np.stack(np.meshgrid(np.arange(16), np.arange(16), indexing="ij"), -1).reshape(-1, 2).astype(float)
    d = hilbert_index(g, bits=4)
    assert sorted(d.tolist()) == list(range(256))
    walk = g[np.argsort(d)]
    assert (np.abs(np.diff(walk, axis=0)).sum(1) == 1).all()
    rng = np.random.default_rng(0)
    p = rng.normal(size=(10000, 2))
    for world in (1, 2, 3, 8):
        parts = partition_hilbert(p, world)
        assert np.array_equal(np.sort(np.concatenate(parts)), np.arange(10000))
        for ix in parts[:-1]:
            assert len(ix) % 256 == 0                       # wave groups never straddle two ranks
        sizes = [len(x) for x in parts]
        assert max(sizes) - min(sizes) <= 512
        # compact: a rank's bounding box area is far below the whole (8 ranks)
        if world == 8:
            area = [np.prod(np.ptp(p[ix], axis=0)) for ix in parts]
            assert np.median(area) < 0.25 * np.prod(np.ptp(p, axis=0))


def test_partition_orb_snaps_the_cut_of_a_symmetric_input_to_the_grid_line():
    from gpu_nbody_simulation_amd.distributed import partition_orb
    rng = np.random.default_rng(3)
    p = rng.uniform(-0.1, 0.1, (20000, 2))
    p[0], p[1] = (-0.1, -0.1), (0.1, 0.1)                    # the box is exactly symmetric: midline at 0
    lo, hi = partition_orb(p, 2)
    ax = 0 if p[lo][:, 0].max() <= 0 or p[hi][:, 0].min() >= 0 else 1
    assert p[lo][:, ax].max() < 0 <= p[hi][:, ax].min()      # no body on the wrong side of the midline
    lo2, hi2 = partition_orb(p, 2, snap=False)
    assert abs(len(lo2) - len(hi2)) <= 1


def test_let_stepper_run_grows_the_blocks_before_they_overflow():
    from dist_standin import LetStandInEngine
    from gpu_nbody_simulation_amd.distributed import LetStepper
    m, p, v = _inputs(200)
    eng = LetStandInEngine()
    eng.upload(p, v, m)
    st = LetStepper(eng, 0, 1, let_cap=64, device=torch.device("cpu"))
    eng.counts_override = 60                                  # pretend the LETs have grown to 60 of 64 quads
    st.run(3, check_every=2)
    assert st.let_cap >= 112 and st.let_cap % 256 == 0        # re-sized to 1.5 x 60 / 0.8, rounded up
    eng.overflow_override = True
    with pytest.raises(RuntimeError):
        st.run(1)
