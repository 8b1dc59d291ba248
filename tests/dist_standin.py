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
        self.spos = torch.zeros(2 * chunk * world, dtype=torch.float32)
        self.svel = torch.zeros(2 * chunk * world, dtype=torch.float32)

    def owned_range(self):
        chunk = ((self.n + self.world - 1) // self.world + 255) // 256 * 256
        return min(self.n, chunk * self.rank), min(self.n, chunk * (self.rank + 1))

    def device_sorted(self):
        return self.spos, self.svel

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
        self.spos[2 * lo:2 * hi] = torch.from_numpy(p.reshape(-1))
        self.svel[2 * lo:2 * hi] = torch.from_numpy(v.reshape(-1))

    def scatter_sorted(self):
        n = self.n
        self.pos[self.perm] = self.spos[:2 * n].numpy().reshape(n, 2)
        self.vel[self.perm] = self.svel[:2 * n].numpy().reshape(n, 2)

    def step(self, k=1):
        for _ in range(k):
            saved = (self.rank, self.world)
            self.rank, self.world = 0, 1
            chunk = self.n
            if self.spos.numel() < 2 * chunk:
                self.spos = torch.zeros(2 * chunk)
                self.svel = torch.zeros(2 * chunk)
            self.step_local()
            self.scatter_sorted()
            self.rank, self.world = saved

    def download(self):
        return self.pos.astype(np.float64), self.vel.astype(np.float64)
