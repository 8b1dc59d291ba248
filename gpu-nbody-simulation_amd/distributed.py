"""One process per GPU over torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in CPU tests).

The reference is single-GPU; this is new design.  Stage 1, implemented here:
every rank holds all bodies and builds the same tree (the build is deterministic, so no tree
exchange is needed), but walks and integrates only its contiguous share of the curve-SORTED (Hilbert order)
bodies -- a compact region of space, so its waves touch few distinct nodes -- and the updated
shares are exchanged with ONE all_gather per step (positions+velocities, 16 B per body in fp32,
fixed-size blocks, in place).  Results are bit-identical to the single-GPU run because a body's
walk does not depend on who executes it.

Stage 2, LetStepper: orthogonal recursive bisection + locally-essential-tree exchange, which
removes the replicated build: every rank holds only its own bodies.

The compute object is injected (`ShardedStepper(engine, ...)`): the product passes a
BarnesHutEngine; the CPU tests pass a stand-in built on the oracle, which lives in tests/ only.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


class _DevPtr:
    """Expose a raw device pointer through __cuda_array_interface__ so torch can wrap it."""

    def __init__(self, ptr: int, nelem: int, typestr: str):
        self.__cuda_array_interface__ = {"shape": (nelem,), "typestr": typestr, "data": (ptr, False),
                                         "version": 3, "strides": None}


def wrap_device_f32(ptr: int, nelem: int, device: torch.device) -> torch.Tensor:
    return torch.as_tensor(_DevPtr(ptr, nelem, "<f4"), device=device)


def wrap_device(ptr: int, nelem: int, typestr: str, device: torch.device) -> torch.Tensor:
    return torch.as_tensor(_DevPtr(ptr, nelem, typestr), device=device)


def hilbert_index(positions, bits: int = 16):
    """Hilbert-curve index (2*bits bits) of 2-D positions on a 2^bits grid over their bounding box
    (the classic xy->d rotation loop, vectorised).  Host, set-up only."""
    import numpy as np
    p = np.asarray(positions, dtype=np.float64)
    if len(p) == 0:
        return np.zeros(0, dtype=np.uint64)
    lo, hi = p.min(0), p.max(0)
    side = (1 << bits) - 1
    q = np.clip(((p - lo) / np.maximum(hi - lo, 1e-300) * side).astype(np.int64), 0, side)
    x, y = q[:, 0].copy(), q[:, 1].copy()
    d = np.zeros(len(p), dtype=np.uint64)
    s = 1 << (bits - 1)
    while s > 0:
        rx = ((x & s) > 0).astype(np.int64)
        ry = ((y & s) > 0).astype(np.int64)
        d += (np.uint64(s) * np.uint64(s)) * ((3 * rx) ^ ry).astype(np.uint64)
        flip = (ry == 0) & (rx == 1)
        x = np.where(flip, s - 1 - x, x)
        y = np.where(flip, s - 1 - y, y)
        swap = ry == 0
        x, y = np.where(swap, y, x), np.where(swap, x, y)
        s >>= 1
    return d


def partition_hilbert(positions, world: int, align: int = 256):
    """Indices of the bodies of every rank: contiguous ranges of the Hilbert order of the positions,
    each returned IN that order, boundaries on multiples of `align` bodies.

    Why a curve range and not orthogonal recursive bisection: the walk's unit is a wave of 64
    Hilbert-consecutive bodies, and its cost is the UNION of their walks.  A rank that owns a curve
    range forms the same groups a single GPU would.  ORB cuts through cells; the few bodies of a cell
    that end up on the far side of a cut form groups strung out along the cut, each costing tens of
    times a compact group, and the kernel waits for them (measured: uniform N = 1M on 2 ranks, ORB
    walk 0.79 ms on the rank with the sliver against 0.18 ms on the other).  The price is that a
    curve range is not a rectangle; bh_let_bounds therefore describes a rank by several boxes."""
    import numpy as np
    order = np.argsort(hilbert_index(positions), kind="stable")
    n = len(order)
    cuts = [0]
    for r in range(1, world):
        c = (n * r // world + align // 2) // align * align
        cuts.append(min(max(c, cuts[-1]), n))
    cuts.append(n)
    return [order[cuts[r]:cuts[r + 1]] for r in range(world)]


def partition_orb(positions, world: int, snap: bool = True):
    """Indices of the bodies of every rank by orthogonal recursive bisection: cut the longer side of
    the bounding box at the weighted median (ranks in proportion, so any world size), recurse.  The
    rank regions are disjoint rectangles, which is what keeps the locally-essential trees small (the
    LET test is against boxes around a peer's bodies); correctness does not depend on it.

    snap: move each cut to the coarsest line of the GLOBAL tree grid (the root box of
    ComputeRootBounds, project.cu:536-573, halved repeatedly) that can be reached by moving at most
    1 % of the piece's bodies across the cut.  A cut that misses a major grid line by a hair leaves the rank on one
    side with a one-body-thick band of bodies in the cells beyond the line; those sort together, form
    64-body groups strung out along the whole cut, and each such group costs tens of ordinary ones
    (measured: uniform N = 1M on 2 ranks, 0.79 ms walk on the rank with the band against 0.18 ms on
    the other).  Symmetric inputs put the median exactly there.  Host side, set-up only."""
    import numpy as np
    p = np.asarray(positions, dtype=np.float64)
    out = [None] * world
    if len(p):
        lo, hi = p.min(0), p.max(0)
        span = max(hi - lo)
        pad = 0.1 * span if span > 0 else 1e-6
        g0, gw = lo - pad, (hi - lo) + 2 * pad            # the root box, per axis

    def split(idx, r0, nr):
        if nr == 1:
            out[r0] = idx
            return
        nl = nr // 2
        if len(idx) == 0:
            split(idx, r0, nl)
            split(idx, r0 + nl, nr - nl)
            return
        q = p[idx]
        ext = q.max(0) - q.min(0)
        ax = int(ext[1] > ext[0])
        k = len(idx) * nl // nr
        order = np.argsort(q[:, ax], kind="stable")
        if snap and 0 < k < len(idx) and ext[ax] > 0:
            c = q[order, ax]
            med = 0.5 * (c[k - 1] + c[k])
            d = max(1, len(idx) // 100)                       # the cut may move past 1 % of the piece's bodies
            c_lo, c_hi = c[max(k - d, 0)], c[min(k + d, len(idx) - 1)]
            for level in range(1, 31):
                w = gw[ax] / (1 << level)
                line = g0[ax] + round((med - g0[ax]) / w) * w
                if c_lo <= line <= c_hi:
                    k = int(np.searchsorted(c, line, side="left"))
                    break
        split(idx[order[:k]], r0, nl)
        split(idx[order[k:]], r0 + nl, nr - nl)

    split(np.arange(len(p)), 0, world)
    return out


def init_process_group_from_env(backend: str | None = None):
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run exports them."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group(backend=backend or ("nccl" if torch.cuda.is_available() else "gloo"),
                                rank=rank, world_size=world)
    return rank, local, world


def _bind_engine_stream(engine, device) -> None:
    """The steppers issue torch collectives on torch's CURRENT stream; the engine enqueues its kernels on
    its own non-blocking stream unless told otherwise, and nothing would order the two (the all_gather
    could read the boxes before let_bounds has written them, the walk could read a block before the
    all_to_all has landed).  Put the engine on torch's stream.  Stand-in engines (CPU tests) have no
    stream."""
    if device is not None and torch.device(device).type == "cuda" and hasattr(engine, "set_stream"):
        engine.set_stream(torch.cuda.current_stream(device).cuda_stream)


class ShardedStepper:
    """step() = local build + walk of the owned sorted range, all_gather, scatter to caller order."""

    def __init__(self, engine, rank: int, world: int, n: int, device: torch.device | None = None,
                 force_exchange: bool = False):
        """force_exchange: take the exchange path even for world == 1 (rehearses the collective on a
        single rank; used by the tests and `bench.py --force-sharded`)."""
        self.eng, self.rank, self.world, self.n = engine, rank, world, n
        self.exchange = world > 1 or force_exchange
        if self.exchange:
            _bind_engine_stream(engine, device)
        per = (n + world - 1) // world
        self.chunk = (per + 255) // 256 * 256       # workgroup-aligned, as bh_owned_range computes it
        engine.set_owned_fraction(rank, world)
        lo, hi = engine.owned_range()
        assert (lo, hi) == (min(n, self.chunk * rank), min(n, self.chunk * (rank + 1)))
        self.lo, self.hi = lo, hi
        if self.exchange:
            sp, sv = engine.device_sorted()
            nel = 2 * self.chunk * world        # float2 per body; buffers hold chunk*world slots
            if isinstance(sp, torch.Tensor):    # stand-in engines hand tensors over directly
                self.spos, self.svel = sp, sv
            else:
                self.spos = wrap_device_f32(sp, nel, device)
                self.svel = wrap_device_f32(sv, nel, device)

    def step(self) -> None:
        if not self.exchange:
            self.eng.step(1)
            return
        self.eng.step_local()
        c2 = 2 * self.chunk
        for buf in (self.spos, self.svel):
            # fixed-size block per rank; the send block is a copy so the collective never aliases
            # its own output (2 MB per rank at N = 1M on 8 ranks)
            mine = buf[self.rank * c2:(self.rank + 1) * c2].clone()
            if dist.is_initialized() and buf.is_cuda and dist.get_backend() == "gloo":
                # rehearsal only (several ranks sharing one GPU, where RCCL cannot run): through the host
                self.eng.sync()
                out = torch.empty(self.world * c2, dtype=buf.dtype)
                dist.all_gather_into_tensor(out, mine.cpu())
                buf[: self.world * c2].copy_(out)
            elif dist.is_initialized():
                dist.all_gather_into_tensor(buf[: self.world * c2], mine)
            else:
                assert self.world == 1
        self.eng.scatter_sorted()


class LetStepper:
    """Distributed step with locally-essential trees (bh_let_* in include/bhgpu.h).

    Every rank holds only its own bodies (engine.upload of its subset: partition_hilbert).  Per
    step: local boxes -> all_gather (256 B per rank) -> local tree under the global box + one
    compact LET per peer -> all_to_all of fixed-size blocks -> forest walk (own tree + received
    LETs) + integrate.  No replicated work, two collectives, no host synchronisation.

    The blocks are fixed-size so that no count has to reach the host inside a step; autotune()
    sizes them once from measured LET sizes, and check() (call it outside timed regions, every so
    often) raises if a LET has outgrown its block since."""

    def __init__(self, engine, rank: int, world: int, let_cap: int, device: torch.device | None = None,
                 ids=None, overlap: bool = False):
        """ids: global identifiers of this rank's bodies in upload order (kept through repartition()).
        overlap: walk the local tree while the LETs are in flight (two walk launches instead of one).
        The second launch costs 15-35 us per step on one MI355X (scripts/let_emulate.py: 0.280 -> 0.301
        ms per rank at N = 1M on 8 ranks); whether the hidden all_to_all is worth more can only be
        measured on a multi-GPU node, so it is off by default."""
        self.eng, self.rank, self.world, self.device = engine, rank, world, device
        _bind_engine_stream(engine, device)
        self.ids = ids
        self.overlap = overlap and hasattr(engine, "let_walk_local")
        # the received blocks must start at the same quad index on every rank (a sender writes links in
        # the receiver's index space) although the ranks' capacities differ: agree on the largest
        fb = torch.tensor([engine.let_local_quads()], dtype=torch.int64)
        if dist.is_initialized() and world > 1:
            if dist.get_backend() != "gloo":
                fb = fb.to(device)
            dist.all_reduce(fb, op=dist.ReduceOp.MAX)
        self.forest_base = int(fb.item())
        self._configure(let_cap)

    def _configure(self, let_cap: int) -> None:
        self.let_cap = let_cap
        self.eng.let_configure(self.rank, self.world, let_cap, self.forest_base)
        lb, ab, sd, rv, nb, k = self.eng.let_pointers()
        if isinstance(lb, torch.Tensor):        # stand-in engines hand tensors over directly
            self.lbounds, self.all_bounds, self.send, self.recv = lb, ab, sd, rv
        else:
            self.lbounds = wrap_device(lb, 4 * k, "<f8", self.device)
            self.all_bounds = wrap_device(ab, 4 * k * self.world, "<f8", self.device)
            self.send = wrap_device(sd, self.world * nb, "|u1", self.device)
            self.recv = wrap_device(rv, self.world * nb, "|u1", self.device)

    def _staged(self) -> bool:
        """Device buffers but a host-only backend (gloo): stage the collectives through the host.  Only
        for rehearsing the multi-rank logic where RCCL cannot run (several ranks on one GPU)."""
        return self.lbounds.is_cuda and dist.is_initialized() and dist.get_backend() == "gloo"

    def _exchange_bounds(self) -> None:
        self.eng.let_bounds()
        if dist.is_initialized() and self.world > 1:
            if self._staged():
                self.eng.sync()
                out = torch.empty(self.all_bounds.numel(), dtype=self.all_bounds.dtype)
                dist.all_gather_into_tensor(out, self.lbounds.cpu())
                self.all_bounds.copy_(out)
            else:
                dist.all_gather_into_tensor(self.all_bounds, self.lbounds)     # (separate buffers: no aliasing)
        else:
            assert self.world == 1
            self.all_bounds.copy_(self.lbounds)

    def step(self, integrate: bool = True) -> None:
        self._exchange_bounds()
        self.eng.let_build()
        if dist.is_initialized() and self._staged():
            self.eng.sync()
            out = torch.empty(self.recv.numel(), dtype=self.recv.dtype)
            dist.all_to_all_single(out, self.send.cpu())
            self.recv.copy_(out)
            if self.overlap:                                  # (nothing to overlap with here; same two launches)
                self.eng.let_walk_local()
                self.eng.let_walk_remote(integrate)
                return
        elif dist.is_initialized() and self.overlap:
            # block q of send -> block rank of q's recv, on the collective's own stream; the local-tree
            # walk does not need it, the second walk launch waits for it
            work = dist.all_to_all_single(self.recv, self.send, async_op=True)
            self.eng.let_walk_local()
            work.wait()
            self.eng.let_walk_remote(integrate)
            return
        elif dist.is_initialized():
            dist.all_to_all_single(self.recv, self.send)
        if integrate:
            self.eng.let_walk()
        else:
            self.eng.let_forces()

    def max_count(self):
        """(largest LET any rank packed in ANY build since the previous call, whether any of those builds
        overflowed -- a LET beyond let_cap or a local tree beyond node_capacity) -- synchronises; the
        all_reduce makes every rank see the same answer, so they raise (or re-size) together."""
        counts, ov = self.eng.let_counts(with_overflow=True)
        t = torch.tensor([max(counts), int(ov)], dtype=torch.int64,
                         device="cpu" if self._staged() else self.lbounds.device)
        if dist.is_initialized() and self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return int(t[0].item()), bool(t[1].item())

    def autotune(self, slack: float = 1.5, floor: int = 256) -> int:
        """One bounds + build on the current state, then let_cap = slack x the largest LET (all ranks
        agree through an all_reduce).  Returns the new let_cap."""
        self._exchange_bounds()
        self.eng.let_build()
        mx, _ = self.max_count()
        cap = max(floor, (int(slack * mx) + 255) // 256 * 256)
        if cap != self.let_cap:
            self._configure(cap)
        return cap

    def check(self) -> int:
        mx, ov = self.max_count()
        if ov:
            raise RuntimeError(f"a locally-essential tree outgrew let_cap={self.let_cap} (largest {mx}); "
                               "results since the last check are invalid -- autotune() again")
        return mx

    def run(self, nsteps: int, check_every: int = 50, grow_at: float = 0.8) -> None:
        """nsteps steps with the block size looked after: every check_every steps the largest LET is
        compared with let_cap (one synchronisation); above grow_at x let_cap the blocks are re-sized
        BEFORE anything overflows, and an overflow that happened anyway raises (see check())."""
        for s in range(nsteps):
            self.step()
            if (s + 1) % check_every == 0 or s + 1 == nsteps:
                mx = self.check()
                if mx > grow_at * self.let_cap:
                    self._configure((int(1.5 * mx / grow_at) + 255) // 256 * 256)

    def repartition(self, partition=partition_orb) -> int:
        """Re-deal the bodies to the ranks (bodies drift; a rank's bodies spread, its boxes overlap its
        neighbours' and the LETs grow).  Set-up-grade: the state goes through the host and the object
        collectives; call it every few hundred steps, not every step.  Needs `ids`; returns the number
        of bodies this rank holds afterwards (the engine's capacity must allow it)."""
        import numpy as np
        if self.ids is None:
            raise ValueError("LetStepper.repartition needs the bodies' global ids (ids=...)")
        pos, vel = self.eng.download()
        mine = (np.asarray(self.ids), pos, vel, self.eng.masses())
        if dist.is_initialized() and self.world > 1:
            pieces = [None] * self.world
            dist.all_gather_object(pieces, mine)
        else:
            pieces = [mine]
        ids = np.concatenate([x[0] for x in pieces])
        order = np.argsort(ids, kind="stable")             # every rank sees the same global arrays
        ids = ids[order]
        pos = np.concatenate([x[1] for x in pieces])[order]
        vel = np.concatenate([x[2] for x in pieces])[order]
        mass = np.concatenate([x[3] for x in pieces])[order]
        sel = partition(pos, self.world)[self.rank]
        self.eng.upload(pos[sel], vel[sel], mass[sel])
        self.ids = ids[sel]
        self.autotune()
        return len(sel)

