"""Multi-GPU exchange logic rehearsed on ONE GPU: two contexts act as ranks 0 and 1 of a world of 2;
their owned sorted slices are copied into each other's exchange buffers (what the RCCL all_gather
does) and scattered back.  The result must equal the single-context run bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import gpu_nbody_simulation_amd as G  # noqa: E402
from gpu_nbody_simulation_amd import initial_conditions as IC  # noqa: E402
from gpu_nbody_simulation_amd.distributed import wrap_device_f32  # noqa: E402


@pytest.mark.parametrize("n,world", [(8192, 2), (10001, 3)])
def test_emulated_ranks_equal_single_context(n, world):
    m, p, v = IC.make("uniform", n, 5)
    cfg = dict(capacity=n, max_depth=16, precision=G.Precision.F32, reference_compat=False)
    dev = torch.device("cuda", 0)
    ref = G.BarnesHutEngine(G.BhConfig(**cfg))
    ref.upload(p, v, m)
    engs = [G.BarnesHutEngine(G.BhConfig(**cfg)) for _ in range(world)]
    chunk = ((n + world - 1) // world + 255) // 256 * 256
    bufs = []
    for r, e in enumerate(engs):
        e.upload(p, v, m)
        e.set_owned_fraction(r, world)
        assert e.owned_range() == (min(n, r * chunk), min(n, (r + 1) * chunk))
        sp, sv = e.device_sorted()
        bufs.append((wrap_device_f32(sp, 2 * chunk * world, dev), wrap_device_f32(sv, 2 * chunk * world, dev)))
    for step in range(3):
        ref.step(1)
        for e in engs:
            e.step_local()
            e.sync()
        for r in range(world):                       # "all_gather": every rank receives every slice
            lo, hi = 2 * r * chunk, 2 * (r + 1) * chunk
            for q in range(world):
                if q != r:
                    bufs[q][0][lo:hi].copy_(bufs[r][0][lo:hi])
                    bufs[q][1][lo:hi].copy_(bufs[r][1][lo:hi])
        torch.cuda.synchronize()
        for e in engs:
            e.scatter_sorted()
            e.sync()
    pr, vr = ref.download()
    for e in engs:
        pe, ve = e.download()
        assert np.array_equal(pe, pr) and np.array_equal(ve, vr)
        e.close()
    ref.close()
