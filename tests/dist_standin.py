"""Stand-in for BarnesHutEngine used ONLY by the CPU multi-process tests: same sharding surface
(set_owned_fraction / owned_range / step_local / device_sorted / scatter_sorted / step), compute
done by the oracle.  It exists so that gpu_nbody_simulation_amd.distributed.ShardedStepper -- the
product's exchange logic -- can run under gloo on a host without a GPU."""
import numpy as np
import torch

from oracle import bh_oracle as O


def _morton_order(pos):
    lo, hi = pos.min(0), pos.max(0)
    q = np.clip(((pos - lo) / np.maximum(hi - lo, 1e-300) * 65535).astype(np.uint64), 0, 65535)
    key = np.zeros(len(pos), dtype=np.uint64)
    for b in range(16):
        key |= ((q[:, 0] >> np.uint64(b)) & np.uint64(1)) << np.uint64(2 * b)
        key |= ((q[:, 1] >> np.uint64(b)) & np.uint64(1)) << np.uint64(2 * b + 1)
    return np.argsort(key, kind="stable")


class OracleStandInEngine:
    def __init__(self, theta=0.5, G=6.67e-11, dt=1.0, max_depth=16):
        self.theta, self.G, self.dt, self.max_depth = theta, G, dt, max_depth
        self.rank, self.world = 0, 1

    def upload(self, pos, vel, mass):
        self.pos = np.array(pos, dtype=np.float32)
        self.vel = np.array(vel, dtype=np.float32)
        self.mass = np.array(mass, dtype=np.float32)
        self.n = len(self.mass)

    def set_owned_fraction(self, rank, world):
        self.rank, self.world = rank, world
        chunk = ((self.n + world - 1) // world + 255) // 256 * 256
        self.sstate = torch.zeros(4 * chunk * world, dtype=torch.float32)

    def owned_range(self):
        chunk = ((self.n + self.world - 1) // self.world + 255) // 256 * 256
        return min(self.n, chunk * self.rank), min(self.n, chunk * (self.rank + 1))

    def device_sorted(self):
        return self.sstate

    def _accel(self, lo, hi):
        p64, m64 = self.pos.astype(np.float64), self.mass.astype(np.float64)
        tree = O.build_tree(p64, m64, self.max_depth)
        self.perm = _morton_order(p64)
        acc = np.zeros((self.n, 2))
        idx = self.perm[lo:hi]
        # the oracle walks a contiguous body range; walk each owned body individually
        for b in idx:
            f = O.compute_forces(tree, p64, m64, theta=self.theta, G=self.G, lo=int(b), hi=int(b) + 1)
            acc[b] = f[b] / m64[b]
        return acc, idx

    def step_local(self):
        lo, hi = self.owned_range()
        acc, idx = self._accel(lo, hi)
        v = self.vel[idx] + (acc[idx] * self.dt).astype(np.float32)
        p = self.pos[idx] + v * np.float32(self.dt)
        self.sstate[4 * lo:4 * hi] = torch.from_numpy(np.concatenate([p, v], axis=1).astype(np.float32).reshape(-1))

    def scatter_sorted(self):
        n = self.n
        st = self.sstate[:4 * n].numpy().reshape(n, 4)
        self.pos[self.perm] = st[:, 0:2]
        self.vel[self.perm] = st[:, 2:4]

    def step(self, k=1):
        for _ in range(k):
            saved = (self.rank, self.world)
            self.rank, self.world = 0, 1
            chunk = self.n
            if self.sstate.numel() < 4 * chunk:
                self.sstate = torch.zeros(4 * chunk)
            self.step_local()
            self.scatter_sorted()
            self.rank, self.world = saved

    def download(self):
        return self.pos.astype(np.float64), self.vel.astype(np.float64)


class LetStandInEngine:
    """Stand-in for the bh_let_* surface of BarnesHutEngine (CPU tests of distributed.LetStepper).

    Its "locally-essential tree" for every peer is simply ALL of its bodies (the theta -> 0 LET),
    written into the fixed-size block for that peer behind a header {sender, destination, count};
    the walk is the oracle's direct sum over own + received bodies.  What this exercises is the
    orchestration: the all_gather of bounds, the block routing of the all_to_all, autotune and the
    overflow check.  The device-side LET logic is covered by tests/test_gpu_let.py."""
    QUAD_BYTES = 80
    BOXES = 8

    def __init__(self, G=6.67e-11, dt=1.0):
        self.G, self.dt = G, dt

    def upload(self, pos, vel, mass):
        self.pos = np.array(pos, dtype=np.float32).reshape(-1, 2)
        self.vel = np.array(vel, dtype=np.float32).reshape(-1, 2)
        self.mass = np.array(mass, dtype=np.float32).reshape(-1)
        self.n = len(self.mass)
        self.gid = np.arange(self.n, dtype=np.int64)
        if not hasattr(self, "capacity"):
            self.capacity = 4 * self.n + 64                   # head-room for migration, as a real context would have

    def download(self):
        return self.pos.astype(np.float64), self.vel.astype(np.float64)

    def masses(self):
        return self.mass.astype(np.float64)

    # ---- migration surface (bh_set_ids ... bh_migrate_unpack), numpy twins of csrc/bh_migrate.hpp
    def set_ids(self, ids):
        self.gid = np.array(ids, dtype=np.int64).reshape(-1)
        assert len(self.gid) == self.n

    def ids(self):
        return self.gid.copy()

    def sync(self):
        pass

    def orb_histogram(self, cuts, level):
        return torch.from_numpy(cuts.histogram(self.pos.astype(np.float64), np.ones(self.n), level).reshape(-1).copy())

    def migrate_pointers(self):
        if not hasattr(self, "_mig"):
            self._mig = (torch.zeros(6 * self.capacity, dtype=torch.float64), torch.zeros(6 * self.capacity, dtype=torch.float64))
        return self._mig[0], self._mig[1], self.capacity

    def migrate_pack(self, cuts):
        own = cuts.owner(self.pos.astype(np.float64))
        order = np.argsort(own, kind="stable")                # grouped by destination, slot order kept
        rec = np.concatenate([self.pos, self.vel, self.mass[:, None], self.gid[:, None].astype(np.float64)], axis=1)
        send = self.migrate_pointers()[0]
        send[: 6 * self.n] = torch.from_numpy(rec[order].astype(np.float64).reshape(-1))
        return [int((own == r).sum()) for r in range(cuts.world)]

    def migrate_unpack(self, n_new):
        rec = self.migrate_pointers()[1][: 6 * n_new].numpy().reshape(n_new, 6)
        self.pos = rec[:, 0:2].astype(np.float32)
        self.vel = rec[:, 2:4].astype(np.float32)
        self.mass = rec[:, 4].astype(np.float32)
        self.gid = rec[:, 5].astype(np.int64)
        self.n = n_new

    def let_local_quads(self):
        if not hasattr(self, "_lq"):
            self._lq = 2 * self.n + 256                       # fixed by the first upload ("capacity"); differs between ranks
        return self._lq

    def let_configure(self, rank, world, let_cap, forest_base):
        assert forest_base >= self.let_local_quads()
        self.forest_base_seen = forest_base
        self.rank, self.world, self.let_cap = rank, world, let_cap
        nb = let_cap * self.QUAD_BYTES
        self.lbounds = torch.zeros(4 * self.BOXES, dtype=torch.float64)
        self.all_bounds = torch.zeros(4 * self.BOXES * world, dtype=torch.float64)
        self.send = torch.zeros(world * nb, dtype=torch.uint8)
        self.recv = torch.zeros(world * nb, dtype=torch.uint8)
        self.counts, self.overflow = [0] * world, False

    def let_pointers(self):
        return self.lbounds, self.all_bounds, self.send, self.recv, self.let_cap * self.QUAD_BYTES, self.BOXES

    def let_bounds(self):
        b = np.tile(np.array([np.inf, -np.inf, np.inf, -np.inf]), (self.BOXES, 1))
        for k in range(self.BOXES):
            sl = self.pos[self.n * k // self.BOXES: self.n * (k + 1) // self.BOXES]
            if len(sl):
                b[k] = [sl[:, 0].min(), sl[:, 0].max(), sl[:, 1].min(), sl[:, 1].max()]
        self.lbounds[:] = torch.from_numpy(b.reshape(-1))

    def let_build(self):
        nb = self.let_cap * self.QUAD_BYTES
        self.seen_bounds = self.all_bounds.numpy().reshape(self.world, self.BOXES, 4).copy()
        rec = np.concatenate([self.pos, self.mass[:, None]], axis=1).astype(np.float32).reshape(-1)
        need = 12 + rec.nbytes
        quads = (need + self.QUAD_BYTES - 1) // self.QUAD_BYTES
        self.overflow = quads > self.let_cap
        send = self.send.numpy()
        for q in range(self.world):
            self.counts[q] = 0 if q == self.rank else quads
            blk = send[q * nb:(q + 1) * nb]
            blk[:] = 0
            cnt = 0 if (self.overflow or q == self.rank) else self.n
            blk[:12] = np.array([self.rank, q, cnt], dtype=np.int32).view(np.uint8)
            blk[12:12 + 12 * cnt] = rec[:3 * cnt].view(np.uint8)

    counts_override = None
    overflow_override = None

    def let_counts(self, with_overflow=False):
        if self.counts_override is not None or self.overflow_override is not None:
            c = [self.counts_override or 0] * self.world
            return (c, bool(self.overflow_override)) if with_overflow else c
        if with_overflow:
            return list(self.counts), self.overflow
        if self.overflow:
            raise RuntimeError("overflow")
        return list(self.counts)

    def _accel(self):
        nb = self.let_cap * self.QUAD_BYTES
        recv = self.recv.numpy()
        ps, ms = [self.pos], [self.mass]
        for r in range(self.world):
            if r == self.rank:
                continue
            blk = recv[r * nb:(r + 1) * nb]
            sender, dest, cnt = blk[:12].view(np.int32)
            assert (sender, dest) == (r, self.rank), f"block {r} of rank {self.rank} came from {sender} for {dest}"
            rec = blk[12:12 + 12 * cnt].view(np.float32).reshape(cnt, 3)
            ps.append(rec[:, :2])
            ms.append(rec[:, 2])
        p = np.concatenate(ps).astype(np.float64)
        m = np.concatenate(ms).astype(np.float64)
        f = O.direct_forces(p, m, G=self.G)
        return (f[:self.n] / m[:self.n, None]) if self.n else np.zeros((0, 2))

    def let_forces(self):
        self.acc = self._accel()

    def let_walk(self):
        a = self._accel()
        self.vel = self.vel + (a * self.dt).astype(np.float32)
        self.pos = self.pos + self.vel * np.float32(self.dt)
