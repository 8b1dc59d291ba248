"""Per-rank device time of the two multi-GPU decompositions, rehearsed on ONE GPU (no collectives):

  replicated : every rank builds the whole tree, walks 1/W of the sorted bodies, scatters
  LET        : every rank builds the tree of its n/W bodies, packs W-1 LETs, walks the forest

W contexts stand in for W ranks; the exchange is done once by device copies, then each rank's
chain (bounds -> build+pack -> forest walk, forces only so the state stays put) is enqueued K times
back to back and timed.  What is missing relative to a real run is the two collectives
(32 B x W all_gather, let_cap x 80 B x W all_to_all).

  python scripts/let_emulate.py [--n 1048576] [--worlds 2,4,8] [--init plummer]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpu_nbody_simulation_amd as G  # noqa: E402
from gpu_nbody_simulation_amd import initial_conditions as IC  # noqa: E402
from gpu_nbody_simulation_amd.distributed import partition_hilbert, partition_orb, wrap_device  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--worlds", default="2,4,8")
    ap.add_argument("--init", default="plummer")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--partition", choices=["hilbert", "orb", "orb-nosnap"], default="orb")
    ap.add_argument("--theta", type=float, default=0.5)
    ap.add_argument("--precision", choices=["f32", "mixed"], default="f32")
    ap.add_argument("--graph", action="store_true", help="also replay each rank's chain from a captured hipGraph")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    n = a.n
    m, p, v = IC.make(a.init, n, 1, quasi_static=True)
    cfg = dict(theta=a.theta, max_depth=21, reference_compat=False,
               precision=G.Precision.MIXED if a.precision == "mixed" else G.Precision.F32)
    res = {"n": n, "init": a.init, "rows": []}

    with G.BarnesHutEngine(G.BhConfig(capacity=n, **cfg)) as e:
        e.upload(p, v, m)
        e.step(3); e.sync()
        t0 = time.perf_counter(); e.step(a.reps); e.sync()
        single = (time.perf_counter() - t0) / a.reps * 1e3
    res["single_gpu_ms"] = single
    print(f"N={n} {a.init}: single context {single:.3f} ms/step", flush=True)

    for world in [int(x) for x in a.worlds.split(",")]:
        # ---- replicated: rank 0's share (all shares are equal-sized; the densest is rank-dependent)
        rep = []
        with G.BarnesHutEngine(G.BhConfig(capacity=n, **cfg)) as e:
            e.upload(p, v, m)
            for r in sorted({0, world // 2, world - 1}):
                # build + walk of the owned range, forces only (integrating without the all_gather
                # would corrupt the other ranks' shares); each call ends in a host sync
                e.set_owned_fraction(r, world)
                for _ in range(3):
                    e._check(e._lib.bh_compute_forces(e._h))
                t0 = time.perf_counter()
                for _ in range(a.reps):
                    e._check(e._lib.bh_compute_forces(e._h))
                rep.append((time.perf_counter() - t0) / a.reps * 1e3)
        # ---- LET
        parts = (partition_hilbert(p, world) if a.partition == "hilbert"
                 else partition_orb(p, world, snap=a.partition == "orb"))
        cap_bodies = max(len(ix) for ix in parts)
        engs, bufs = [], []
        fb = None

        def wire(e):
            lb, ab, sd, rv, nb, k = e.let_pointers()
            return (wrap_device(lb, 4 * k, "<f8", dev), wrap_device(ab, 4 * k * world, "<f8", dev),
                    wrap_device(sd, world * nb, "|u1", dev), wrap_device(rv, world * nb, "|u1", dev), nb)

        for r, ix in enumerate(parts):
            e = G.BarnesHutEngine(G.BhConfig(capacity=max(len(ix), 1), **cfg))     # sized for its own bodies
            e.upload(p[ix], v[ix], m[ix])
            engs.append(e)
        fb = max(e.let_local_quads() for e in engs)                               # the agreed forest_base
        for r, e in enumerate(engs):
            e.let_configure(r, world, 1 << 16, fb)
            bufs.append(wire(e))

        def exchange_bounds():
            for e in engs:
                e.let_bounds(); e.sync()
            allb = torch.cat([b[0] for b in bufs])
            for b in bufs:
                b[1].copy_(allb)
            torch.cuda.synchronize()

        def exchange_lets():
            for r in range(world):
                nb = bufs[r][4]
                for q in range(world):
                    if q != r:
                        bufs[q][3][r * nb:(r + 1) * nb].copy_(bufs[r][2][q * nb:(q + 1) * nb])
            torch.cuda.synchronize()

        exchange_bounds()
        for e in engs:
            e.let_build(); e.sync()
        counts = np.array([e.let_counts() for e in engs])
        cap = max(256, (int(1.5 * counts.max()) + 255) // 256 * 256)
        for i, e in enumerate(engs):
            e.let_configure(i, world, cap, fb); bufs[i] = wire(e)
        exchange_bounds()
        for e in engs:
            e.let_build(); e.sync()
        exchange_lets()
        let, let2 = [], []
        for e in engs:
            for two, acc in ((False, let), (True, let2)):
                def chain():
                    e.let_bounds(); e.let_build()
                    if two:
                        e.let_walk_local(); e.let_walk_remote(False)
                    else:
                        e.let_forces()
                for _ in range(3):
                    chain()
                e.sync()
                t0 = time.perf_counter()
                for _ in range(a.reps):
                    chain()
                e.sync()
                acc.append((time.perf_counter() - t0) / a.reps * 1e3)
        # the same chain replayed from a hipGraph (one launch of ~14 kernel nodes instead of ~14 launches): does the
        # host-side launch path or the GPU-side boundary between dependent kernels set a rank's step time?
        letg = []
        if a.graph:
            for e in engs:
                st = torch.cuda.Stream()
                e.sync()
                e.set_stream(st.cuda_stream)
                g = torch.cuda.CUDAGraph()
                try:
                    with torch.cuda.graph(g, stream=st, capture_error_mode="relaxed"):
                        e.let_bounds(); e.let_build(); e.let_forces()
                    for _ in range(3):
                        g.replay()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(a.reps):
                        g.replay()
                    torch.cuda.synchronize()
                    letg.append((time.perf_counter() - t0) / a.reps * 1e3)
                except Exception as ex:                       # noqa: BLE001 -- report, the eager numbers stand
                    print("graph capture failed:", repr(ex)[:200], flush=True)
                    break
        for e in engs:
            e.close()
        row = {"world": world, "replicated_ms_max": max(rep), "let_ms_max": max(let), "let_ms_mean": float(np.mean(let)),
               "let_graph_ms_max": max(letg) if len(letg) == len(engs) else None,
               "let_two_launch_ms_max": max(let2), "let_two_launch_ms_mean": float(np.mean(let2)),
               "let_quads_mean": float(counts[counts > 0].mean()), "let_quads_max": int(counts.max()),
               "let_cap": cap, "all_to_all_bytes_per_rank": cap * 80 * (world - 1),
               "bodies_per_rank": cap_bodies}
        res["rows"].append(row)
        print(json.dumps(row), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    res["partition"] = a.partition
    with open(f"gpurun_out/let_emulate_{a.init}_{n}_{a.partition}.json", "w") as fh:
        json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
