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


# ---- orthogonal recursive bisection as a cut tree (device-side migration) -----------------
ORB_BINS = 4096          # = BH_ORB_BINS: histogram bins across the root box along one axis (depth-12 grid lines)


class OrbCuts:
    """The ownership rule of the LET decomposition: a binary tree of axis-aligned cuts over the global root
    box.  The node that splits the ranks [r0, r0 + nr) (nr > 1) into nl = nr // 2 and nr - nl sends a body
    with coordinate[axis] < value to the left; cuts are stored in pre-order (the left subtree's nl - 1 cuts
    follow their parent directly, the right subtree's come after those).  Mirrors bh_orb_cuts of
    include/bhgpu.h."""

    def __init__(self, world: int, box):
        import numpy as np
        self.world = world
        self.box = np.asarray(box, dtype=np.float64).copy()
        self.axis = np.zeros(max(world - 1, 0), dtype=np.int32)
        self.value = np.zeros(max(world - 1, 0), dtype=np.float64)

    def regions(self, level: int):
        """[(cut index, first rank, number of ranks, box)] of the tree nodes at depth `level` that still
        have to be split (nr > 1), boxes narrowed by the cuts above them."""
        out = []

        def walk(k, r0, nr, box, d):
            if nr <= 1:
                return
            if d == level:
                out.append((k, r0, nr, box))
                return
            nl = nr // 2
            ax = int(self.axis[k])
            left, right = box.copy(), box.copy()
            left[2 * ax + 1] = self.value[k]
            right[2 * ax] = self.value[k]
            walk(k + 1, r0, nl, left, d + 1)
            walk(k + nl, r0 + nl, nr - nl, right, d + 1)

        walk(0, 0, self.world, self.box.copy(), 0)
        return out

    def depth(self) -> int:
        d, w = 0, 1
        while w < self.world:
            w *= 2
            d += 1
        return d

    def descend(self, pos, max_level: int = 64):
        """(first rank, cut index, number of ranks) per body after at most max_level cuts: the numpy twin of
        orb_descend in csrc/bh_migrate.hpp."""
        import numpy as np
        p = np.asarray(pos, dtype=np.float64).reshape(-1, 2)
        r0 = np.zeros(len(p), dtype=np.int64)
        nr = np.full(len(p), self.world, dtype=np.int64)
        k = np.zeros(len(p), dtype=np.int64)
        for _ in range(min(max_level, self.depth())):
            act = nr > 1
            if not act.any():
                break
            kk = np.where(act, k, 0)
            nl = nr // 2
            v = np.where(self.axis[kk] == 1, p[:, 1], p[:, 0]) if len(self.axis) else p[:, 0]
            left = act & (v < (self.value[kk] if len(self.value) else 0.0))
            right = act & ~left
            k = np.where(left, k + 1, np.where(right, k + nl, k))
            r0 = np.where(right, r0 + nl, r0)
            nr = np.where(left, nl, np.where(right, nr - nl, nr))
        return r0, k, nr

    def owner(self, pos):
        return self.descend(pos)[0]

    def histogram(self, pos, weights, level: int):
        """Weighted histograms of the regions at `level` (numpy twin of orb_hist_kernel): int64 array
        [max(world - 1, 1), ORB_BINS], row k = the region whose cut is k."""
        import numpy as np
        p = np.asarray(pos, dtype=np.float64).reshape(-1, 2)
        h = np.zeros((max(self.world - 1, 1), ORB_BINS), dtype=np.int64)
        if len(p) == 0:
            return h
        _, k, nr = self.descend(p, level)
        act = nr > 1
        ax = self.axis[np.where(act, k, 0)] if len(self.axis) else np.zeros(len(p), dtype=np.int32)
        lo = np.where(ax == 1, self.box[2], self.box[0])
        hi = np.where(ax == 1, self.box[3], self.box[1])
        t = (np.where(ax == 1, p[:, 1], p[:, 0]) - lo) / (hi - lo) * ORB_BINS
        b = np.where(t >= 0, np.where(t < ORB_BINS, t, ORB_BINS - 1), 0)
        b = np.nan_to_num(b, nan=0.0).astype(np.int64)
        w = np.maximum(np.asarray(weights, dtype=np.int64), 1)
        np.add.at(h, (k[act], b[act]), w[act])
        return h


def choose_cut(hist_row, region_box, root_box, axis: int, frac: float, tol: float = 0.01):
    """Cut coordinate for one region from its all-reduced histogram: the bin edge where the cumulative
    weight is closest to frac x total, then moved to the COARSEST line of the tree grid (the edge index
    with the most factors of two) that keeps the left weight within tol x total of it.  Every rank computes
    the same value from the same integers.  (A cut that misses a major grid line by a hair leaves one rank
    with a one-body-thick band of bodies beyond the line, whose 64-body groups are strung out along the
    whole cut -- measured in round 1: 0.79 ms against 0.18 ms for the walk of that rank.)"""
    import numpy as np
    h = np.asarray(hist_row, dtype=np.float64)
    lo, hi = root_box[2 * axis], root_box[2 * axis + 1]
    width = (hi - lo) / ORB_BINS
    # edges strictly inside the region
    e_lo = int(np.floor((region_box[2 * axis] - lo) / width + 1e-9)) + 1
    e_hi = int(np.ceil((region_box[2 * axis + 1] - lo) / width - 1e-9)) - 1
    e_lo, e_hi = max(e_lo, 1), min(e_hi, ORB_BINS - 1)
    total = h.sum()
    if e_hi < e_lo:
        return 0.5 * (region_box[2 * axis] + region_box[2 * axis + 1])
    if total <= 0:
        e = (e_lo + e_hi) // 2
        return lo + e * width
    cum = np.cumsum(h)                                   # cum[e - 1] = weight left of edge e
    edges = np.arange(e_lo, e_hi + 1)
    err = np.abs(cum[edges - 1] - frac * total)
    best = float(err.min())
    ok = edges[err <= best + tol * total]
    tz = np.array([(int(e) & -int(e)).bit_length() for e in ok])      # trailing zeros + 1: coarseness of the grid line
    cand = ok[tz == tz.max()]
    e = int(cand[np.argmin(np.abs(cum[cand - 1] - frac * total))])
    return lo + e * width


def padded_root_box(xmin, xmax, ymin, ymax):
    """ComputeRootBounds (project.cu:553-570): 10 % of the larger extent on every side."""
    span = max(xmax - xmin, ymax - ymin)
    pad = 0.1 * span if span > 0 else 1e-6
    return [xmin - pad, xmax + pad, ymin - pad, ymax + pad]



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
    """step() = local build + walk of the owned sorted range, ONE in-place all_gather of {x, y, vx, vy},
    scatter to caller order."""

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
            st = engine.device_sorted()
            nel = 4 * self.chunk * world        # {x, y, vx, vy} per body; the buffer holds chunk*world slots
            # stand-in engines hand a tensor over directly
            self.sstate = st if isinstance(st, torch.Tensor) else wrap_device_f32(st, nel, device)

    def step(self) -> None:
        if not self.exchange:
            self.eng.step(1)
            return
        self.eng.step_local()
        c4 = 4 * self.chunk
        buf = self.sstate
        mine = buf[self.rank * c4:(self.rank + 1) * c4]
        if dist.is_initialized() and buf.is_cuda and dist.get_backend() == "gloo":
            # rehearsal only (several ranks sharing one GPU, where RCCL cannot run): through the host
            self.eng.sync()
            out = torch.empty(self.world * c4, dtype=buf.dtype)
            dist.all_gather_into_tensor(out, mine.cpu())
            buf[: self.world * c4].copy_(out)
        elif dist.is_initialized() and not buf.is_cuda:
            dist.all_gather_into_tensor(buf[: self.world * c4], mine.clone())      # gloo: no in-place form
        elif dist.is_initialized():
            # ONE collective per step, in place: rank r's block already sits at block r of the buffer
            # (the in-place form of ncclAllGather: sendbuff == recvbuff + rank * count), 16 B per body
            dist.all_gather_into_tensor(buf[: self.world * c4], mine)
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
        if ids is not None:
            engine.set_ids(ids)                       # the engine owns the ids: they migrate with the bodies
        self.cuts = None                              # OrbCuts of the last rebalance()
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
        # profile = True: step() brackets its phases with events on the current stream (phase_ms() reads them)
        self.profile = False
        self._marks = []

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

    def _mark(self, marks) -> None:
        if marks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append(ev)

    def phase_ms(self) -> dict:
        """Mean milliseconds per phase over the profiled steps since the last call (profile = True): bounds +
        all_gather, local build + LET extraction, all_to_all, walk -- or, with overlap, the local walk under the
        all_to_all, the part of the all_to_all that stayed exposed, and the remote walk.  Synchronises."""
        torch.cuda.synchronize()
        names = (["bounds_allgather", "build_let", "walk_local", "all_to_all_exposed", "walk_remote"] if self.overlap
                 else ["bounds_allgather", "build_let", "all_to_all", "walk"])
        acc = {k: 0.0 for k in names}
        steps = [m for m in self._marks if len(m) == len(names) + 1]
        for m in steps:
            for k, (a, b) in zip(names, zip(m[:-1], m[1:])):
                acc[k] += a.elapsed_time(b)
        self._marks = []
        out = {k: v / max(len(steps), 1) for k, v in acc.items()}
        out["steps_profiled"] = len(steps)
        return out

    def step(self, integrate: bool = True) -> None:
        marks = [] if (self.profile and self.lbounds.is_cuda) else None
        if marks is not None:
            self._marks.append(marks)
        self._mark(marks)
        self._exchange_bounds()
        self._mark(marks)
        self.eng.let_build()
        self._mark(marks)
        if dist.is_initialized() and self._staged():
            self.eng.sync()
            out = torch.empty(self.recv.numel(), dtype=self.recv.dtype)
            dist.all_to_all_single(out, self.send.cpu())
            self.recv.copy_(out)
            if self.overlap:                                  # (nothing to overlap with here; same two launches)
                self.eng.let_walk_local()
                self._mark(marks)
                self._mark(marks)
                self.eng.let_walk_remote(integrate)
                self._mark(marks)
                return
        elif dist.is_initialized() and self.overlap:
            # block q of send -> block rank of q's recv, on the collective's own stream; the local-tree
            # walk does not need it, the second walk launch waits for it
            work = dist.all_to_all_single(self.recv, self.send, async_op=True)
            self.eng.let_walk_local()
            self._mark(marks)
            work.wait()
            self._mark(marks)
            self.eng.let_walk_remote(integrate)
            self._mark(marks)
            return
        elif dist.is_initialized():
            dist.all_to_all_single(self.recv, self.send)
        if self.overlap:                                      # (no process group: world 1)
            self.eng.let_walk_local()
            self._mark(marks)
            self._mark(marks)
            self.eng.let_walk_remote(integrate)
            self._mark(marks)
            return
        self._mark(marks)
        if integrate:
            self.eng.let_walk()
        else:
            self.eng.let_forces()
        self._mark(marks)

    @property
    def ids(self):
        """Global identifiers of this rank's bodies in the order download() returns them."""
        return self.eng.ids()

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

    def run(self, nsteps: int, check_every: int = 50, grow_at: float = 0.8, rebalance_every: int = 0) -> None:
        """nsteps steps with the block size looked after: every check_every steps the largest LET is
        compared with let_cap (one synchronisation); above grow_at x let_cap the blocks are re-sized
        BEFORE anything overflows, and an overflow that happened anyway raises (see check()).
        rebalance_every > 0: every so many steps the bodies are re-dealt on the device (rebalance())."""
        for s in range(nsteps):
            self.step()
            if rebalance_every and (s + 1) % rebalance_every == 0 and s + 1 < nsteps:
                self.check()
                self.rebalance()
            elif (s + 1) % check_every == 0 or s + 1 == nsteps:
                mx = self.check()
                if mx > grow_at * self.let_cap:
                    self._configure((int(1.5 * mx / grow_at) + 255) // 256 * 256)

    # -- device-side migration and re-balancing ---------------------------------------------------------
    def _host(self, t: torch.Tensor) -> torch.Tensor:
        """A tensor the collective can take: device tensors as they are on RCCL, host copies on gloo."""
        if t.is_cuda and dist.is_initialized() and dist.get_backend() == "gloo":
            self.eng.sync()
            return t.cpu()
        return t

    def _all_reduce_(self, t: torch.Tensor, op) -> torch.Tensor:
        if dist.is_initialized() and self.world > 1:
            h = self._host(t)
            dist.all_reduce(h, op=op)
            if h is not t:
                t.copy_(h)
        return t

    def rebalance(self, tol: float = 0.01) -> int:
        """Re-derive the ORB cuts from a distributed weighted histogram and move the bodies that are on the
        wrong side -- all on the devices: no rank ever holds more than its own bodies, nothing is pickled.

          1. global root box: all_gather of the ranks' boxes (the step's own collective), padded as
             ComputeRootBounds does (project.cu:553-570);
          2. level by level (ceil(log2 W) rounds): every region's axis = the longer side of its box; the
             engine histograms its bodies over ORB_BINS bins across the root box along that axis, each body
             weighted by the cost its 64-body group had in the last walk; one all_reduce(SUM) of the integer
             histograms; every rank picks the same cuts from the same integers (choose_cut: the weighted
             median, snapped to the coarsest tree-grid line within `tol` of it);
          3. the engine classifies its bodies by the cut tree and groups them by destination (stable);
             all_to_all of the W counts; ONE all_to_all_single of the 48-byte records with those splits,
             device pointer to device pointer; the received records become the local state;
          4. autotune(): the LET blocks are re-sized for the new domains.
        Returns the number of bodies this rank holds afterwards."""
        import numpy as np
        W = self.world
        # 1. root box
        self._exchange_bounds()
        ab = self.all_bounds
        if ab.is_cuda:
            self.eng.sync()
        b = ab.cpu().numpy().reshape(-1, 4)
        b = b[np.isfinite(b).all(1) & (b[:, 0] <= b[:, 1])]
        box = padded_root_box(b[:, 0].min(), b[:, 1].max(), b[:, 2].min(), b[:, 3].max()) if len(b) else [0, 1, 0, 1]
        cuts = OrbCuts(W, box)
        # 2. cuts, one level per round
        for level in range(cuts.depth()):
            regs = cuts.regions(level)
            for k, _, _, rb in regs:
                cuts.axis[k] = int((rb[3] - rb[2]) > (rb[1] - rb[0]))
            h = self.eng.orb_histogram(cuts, level)
            if not isinstance(h, torch.Tensor):
                h = wrap_device(h[0], h[1], "<i8", self.device)
            h = self._all_reduce_(h, dist.ReduceOp.SUM)
            hh = h.cpu().numpy().reshape(-1, ORB_BINS)
            for k, _, nr, rb in regs:
                cuts.value[k] = choose_cut(hh[k], rb, cuts.box, int(cuts.axis[k]), (nr // 2) / nr, tol)
        # 3. migration
        send_counts = self.eng.migrate_pack(cuts)
        n_old = int(sum(send_counts))
        sc = torch.tensor(send_counts, dtype=torch.int64)
        rc = torch.empty(W, dtype=torch.int64)
        on_dev = dist.is_initialized() and W > 1 and dist.get_backend() != "gloo"
        if dist.is_initialized() and W > 1:
            if on_dev:
                sc, rc = sc.to(self.device), rc.to(self.device)
            dist.all_to_all_single(rc, sc)
            sc, rc = sc.cpu(), rc.cpu()
        else:
            rc = sc.clone()
        recv_counts = [int(x) for x in rc]
        n_new = sum(recv_counts)
        sp, rp, cap = self.eng.migrate_pointers()
        too_many = torch.tensor([1 if n_new > cap else 0], dtype=torch.int64, device=self.device if on_dev else "cpu")
        self._all_reduce_(too_many, dist.ReduceOp.MAX)
        if int(too_many.item()):
            raise RuntimeError(f"rebalance: a rank would receive more bodies than its capacity ({n_new} > {cap} here); "
                               "create the contexts with head-room")
        rec = 6
        if isinstance(sp, torch.Tensor):
            send, recv = sp, rp
        else:
            send = wrap_device(sp, cap * rec, "<f8", self.device)
            recv = wrap_device(rp, cap * rec, "<f8", self.device)
        if dist.is_initialized() and W > 1:
            ins, outs = [c * rec for c in send_counts], [c * rec for c in recv_counts]
            if send.is_cuda and not on_dev:                     # rehearsal on gloo: through the host
                self.eng.sync()
                out = torch.empty(n_new * rec, dtype=torch.float64)
                dist.all_to_all_single(out, send[: n_old * rec].cpu(), outs, ins)
                recv[: n_new * rec].copy_(out)
            else:
                dist.all_to_all_single(recv[: n_new * rec], send[: n_old * rec], outs, ins)
        else:
            recv[: n_new * rec].copy_(send[: n_old * rec])
        self.eng.migrate_unpack(n_new)
        self.cuts = cuts
        # 4. the LET blocks for the new domains
        self.autotune()
        return n_new
