"""One process per GPU over torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in CPU tests).

The reference is single-GPU; this is new design.  Stage 1, implemented here:
every rank holds all bodies and builds the same tree (the build is deterministic, so no tree
exchange is needed), but walks and integrates only its contiguous share of the MORTON-SORTED
bodies -- a compact region of space, so its waves touch few distinct nodes -- and the updated
shares are exchanged with ONE all_gather per step (positions+velocities, 16 B per body in fp32,
fixed-size blocks, in place).  Results are bit-identical to the single-GPU run because a body's
walk does not depend on who executes it.

The orthogonal-recursive-bisection + locally-essential-tree exchange that removes the replicated
build is the next step on this path and is described in DESIGN.md.

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


class ShardedStepper:
    """step() = local build + walk of the owned sorted range, all_gather, scatter to caller order."""

    def __init__(self, engine, rank: int, world: int, n: int, device: torch.device | None = None,
                 force_exchange: bool = False):
        """force_exchange: take the exchange path even for world == 1 (rehearses the collective on a
        single rank; used by the tests and `bench.py --force-sharded`)."""
        self.eng, self.rank, self.world, self.n = engine, rank, world, n
        self.exchange = world > 1 or force_exchange
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
            if dist.is_initialized():
                dist.all_gather_into_tensor(buf[: self.world * c2], mine)
            else:
                assert self.world == 1
        self.eng.scatter_sorted()
