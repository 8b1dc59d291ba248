#!/usr/bin/env python3
"""bench.py -- body-steps/s of the Barnes-Hut step on MI355X (contract: see the task statement).

    python bench.py                       # N=1 GPU, defaults finish in a couple of minutes
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over all bodies: root box -> keys -> radix sort -> tree
nodes/COM -> theta-walk -> integrate, all on the device, bodies resident in HBM before the timed
region.  Workload = BASELINE.json's metric configuration: N = 1,048,576 bodies, theta = 0.5,
Plummer sphere (configs[2]), fp32.  With --gpus N the SAME bodies are split over N ranks
(strong scaling), one all_gather per step over RCCL.

One JSON line on stdout (rank 0).  Besides the contract's fields it carries
  roofline     : the walk kernel's algorithmic bytes / its HIP-event time vs the 8 TB/s HBM peak
  cpu_baseline : the oracle (a port of the reference's CPU path) timed on this host, 1 core,
                 one full step of the same workload
  minteractions_per_s : the metric's second half (accepted body-node force evaluations / s)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
NODE_BYTES = 20         # fp32 node: one quarter of the 80-byte sibling-quad record (bh_nodes.hpp QuadF)
# Issue-rate roofline of the walk.  Cycles of one SIMD per wave-instruction, measured
# with 8 waves per SIMD by scripts/calib/issue_calib.hip (profiles/r02_final/issue_calib.txt): plain fp32
# VALU 2.2; VALU with an SGPR operand, v_pk_*_f32, v_cmp/v_cmpx, v_writelane, v_readfirstlane 4.1-4.3;
# v_rsq_f32 8.3.  Per evaluated child the loop issues v_pk_add(s) + v_mul + v_fmac + v_cmpx + v_rsq +
# v_mul(s) + 2 v_mul + 2 v_fmac; per quad the stack traffic.  Round 3's loop keeps the first opened child of a
# quad in scalar registers (no push, no pop): PMC counts 169.5 M vector instructions per launch for 14.73 M
# children and 4.43 M quads (profiles/r03_final/pmc_summary.csv), i.e. 10 per child and 5.0 per quad -- 2.5
# v_readlane + 2.5 v_writelane on average where round 2 had 3 + 3.
VALU_CYCLES_PER_CHILD = 4.2 + 2.2 + 2.2 + 4.3 + 8.3 + 4.2 + 2.2 + 2.2 + 2.2 + 2.2
VALU_CYCLES_PER_QUAD = 2.5 * 4.1 + 2.5 * 4.2
N_SIMDS, SHADER_CLOCK_HZ = 1024, 2.4e9
# BH_PRECISION_F64 walk (csrc/bh_walk_f64.hpp, walk64_asm).  Cycles of one SIMD per wave-instruction with 8 resident waves,
# scripts/calib/f64_issue_calib.hip: any fp64 VALU instruction 5.3, v_rsq_f64 17, v_cmpx 4.3-5.3, lane moves 4.1-4.2.
# Per evaluated node: dx, dy, d2 (4 x fp64) + the compare; per node some lane accepts: v_rsq_f64 + 4 (Newton) + 5 (weight)
# + 2 (sums); per quad: on average 2.5 lane reads + 2.5 lane writes of the stack like the fp32 loop (the first opened child
# stays in scalar registers).  Algorithmic bytes: 40 B per evaluated node (32-byte node + 8-byte link) + 92 B per body
# (position 16, index 4, mass 8, velocity r/w 32, position w 16, force w 16).
F64_CYCLES_PER_NODE = 4 * 5.3 + 5.3
F64_CYCLES_PER_ACCEPTED = 17.0 + 11 * 5.3
F64_CYCLES_PER_QUAD = 2.5 * 4.1 + 2.5 * 4.2
F64_NODE_BYTES, F64_BODY_BYTES = 40, 92
# BH_PRECISION_F64_EXACT walk (csrc/bh_walk_exact.hpp, walk_exact_asm), same calibrated costs.  Per evaluated (non-empty)
# node: dx, dy, dx^2, dy^2, d2 + the threshold compare; per node some lane takes a term from: two range compares, G m_i m,
# v_rsq_f64 + two v_rcp_f64, 30 more fp64 instructions (Newton steps, quotient corrections, products), two sums.
EXACT_CYCLES_PER_NODE = 6 * 5.3
EXACT_CYCLES_PER_ACCEPTED = 3 * 17.0 + (2 + 1 + 30 + 2) * 5.3
# algorithmic bytes of one whole step per body, fp32 state -- fixed pipeline ~270 B + walk 12 + 20 U64 + 8
SURVEY_PIPELINE_BYTES, SURVEY_WALK_FIXED_BYTES = 270, 20


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n-bodies", type=int, default=1 << 20)
    ap.add_argument("--init", choices=["plummer", "uniform"], default="plummer")
    ap.add_argument("--theta", type=float, default=0.5)
    ap.add_argument("--max-depth", type=int, default=21)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--precision", choices=["f32", "mixed"], default="f32",
                    help="f32: fp32 state and forces (the headline configuration); mixed: fp64 state, fp32 "
                         "forces (BASELINE config 'fp64 positions / fp32 forces')")
    ap.add_argument("--lds-stack", action="store_true", help="A/B: LDS traversal stack variant")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the extra legs (second distribution, dynamic workload, exact fp64 mode)")
    ap.add_argument("--drift-cells", type=float, default=1.0,
                    help="dynamic leg: every body drifts this many depth-12 cell widths per step")
    ap.add_argument("--decomposition", choices=["let", "replicated"], default="let",
                    help="multi-GPU scheme (DESIGN.md 9): 'let' = ORB partition, local trees, locally-"
                         "essential-tree exchange (bodies live on one rank only); 'replicated' = every rank "
                         "builds the whole tree and walks a share, one all_gather per step")
    ap.add_argument("--let-overlap", choices=["auto", "on", "off"], default="auto",
                    help="LET decomposition: walk the local tree while the all_to_all is in flight (two walk "
                         "launches instead of one).  auto: from 4 ranks up, where the all_to_all (3+ peers) is "
                         "expected to cost more than the 15-20 us of the second launch; unmeasured on this "
                         "1-GPU development box")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo: rehearsal only (collectives staged through the host; lets several ranks "
                         "share one GPU together with BHGPU_REHEARSE_ON_DEVICE)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="take the multi-GPU exchange path (step_local, all_gather, scatter) even on one rank")
    ap.add_argument("--other-configs", default="C2,C4,C5",
                    help="comma-separated BASELINE configurations for the other_configs leg ('' = none)")
    ap.add_argument("--cpu-sample", type=int, default=0,
                    help="bodies walked by the CPU baseline (0 = all, i.e. one full step)")
    ap.add_argument("--dry-run", action="store_true",
                    help="plumbing check: parse, (spawn,) join the process group, all-reduce the ranks, print what a real "
                         "run would be launched as -- no device is touched (tests/test_bench_launch_cpu.py)")
    return ap.parse_args()


def spawn_ranks(a, argv):
    """`bench.py --gpus N` started WITHOUT torch.distributed.run (no RANK in the environment): start the N ranks ourselves --
    a child `python -m torch.distributed.run ... bench.py <the same arguments>` -- relay its output and leave with its code.
    (this used to print a note on stderr and measure ONE GPU.)  Called before torch is imported: no device has
    been touched by this process, and the children are fresh processes, not an exec of this one."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print("bench.py: --gpus %d without a launcher; starting the ranks: %s" % (a.gpus, " ".join(cmd)), file=sys.stderr)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def cpu_baseline(mass, pos, theta, sample):
    """The oracle (port of project.cu:575-675 / main_approach_2.cpp, uncapped tree) on ONE core:
    one tree build over all bodies + the walk of `sample` bodies, scaled to body-steps/s."""
    from oracle import bh_oracle as O
    n = len(mass)
    sample = n if sample <= 0 else min(sample, n)
    t0 = time.perf_counter()
    tree = O.build_tree(pos, mass, 0)
    t1 = time.perf_counter()
    f, st = O.compute_forces(tree, pos, mass, theta=theta, compat_self_skip=False, lo=0, hi=sample,
                             with_stats=True)
    t2 = time.perf_counter()
    build_s, walk_s = t1 - t0, t2 - t1
    est_step = build_s + walk_s * n / sample
    model = ""
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    # also the reference's own depth cap (QUADTREE_MAX_DEPTH = 10, project.cu:61),
    # which is degenerate at this size (a saturated tree, ~4 bodies per leaf cell, aggregated)
    t3 = time.perf_counter()
    tree10 = O.build_tree(pos, mass, 10)
    t4 = time.perf_counter()
    s10 = max(1, sample // 8)
    O.compute_forces(tree10, pos, mass, theta=theta, compat_self_skip=True, lo=0, hi=s10)
    t5 = time.perf_counter()
    cap10 = n / ((t4 - t3) + (t5 - t4) * n / s10)
    return {
        "value": n / est_step, "unit": "body-steps/s", "cores": 1, "kind": "port",
        "depth_cap_10_body_steps_per_s": cap10,
        "sample": (f"one step at N={n}: full tree build ({build_s:.2f} s) + theta-walk of {sample} bodies "
                   f"({walk_s:.2f} s)" + ("" if sample == n else ", walk scaled to N")),
        "force_update_only_body_steps_per_s": sample / walk_s,
        "minteractions_per_s": st.interactions / walk_s / 1e6,
        "interactions_per_body": st.interactions / sample,
        "cpu_model": model,
    }


def _lib_is_product():
    from gpu_nbody_simulation_amd import _lib
    return _lib.is_product_library()


def _digest():
    from gpu_nbody_simulation_amd.build import source_digest
    return source_digest()


def rank_churn(p0, p1, depth=16):
    """Fraction of bodies whose rank in the space-filling-curve order, and whose depth-12 cell, changed
    between two consecutive states (host-side, untimed; Morton order on the first state's root box)."""
    lo, hi = p0.min(0), p0.max(0)
    span = float(max(hi - lo))
    pad = 0.1 * span if span > 0 else 1e-6
    org, width = lo - pad, (hi - lo) + 2 * pad

    def keys(p):
        q = np.clip(((p - org) / width * (1 << depth)).astype(np.int64), 0, (1 << depth) - 1)
        k = np.zeros(len(p), dtype=np.int64)
        for b in range(depth):
            k |= ((q[:, 0] >> b) & 1) << (2 * b)
            k |= ((q[:, 1] >> b) & 1) << (2 * b + 1)
        return k
    k0, k1 = keys(p0), keys(p1)
    r0 = np.empty(len(p0), dtype=np.int64)
    r1 = np.empty(len(p0), dtype=np.int64)
    r0[np.argsort(k0, kind="stable")] = np.arange(len(p0))
    r1[np.argsort(k1, kind="stable")] = np.arange(len(p0))
    sh = 2 * (depth - 12)
    return {"rank_changed_frac": float((r0 != r1).mean()), "cell12_changed_frac": float(((k0 >> sh) != (k1 >> sh)).mean()),
            "median_abs_rank_shift": float(np.median(np.abs(r1 - r0)))}


def _single_gpu_file(a, n):
    import tempfile
    return os.path.join(tempfile.gettempdir(), f"bhgpu_bench_1gpu_{_digest()}_{a.init}_{n}_{a.theta}_{a.precision}.json")


def single_gpu_record(a, n, write=None):
    """The `--gpus 1` leg of the same library and workload, kept between the runs of one scaling series (the driver runs
    N = 1, 2, 4, 8 back to back on one node): written by a one-GPU run, read by the others.  None if there is none."""
    f = _single_gpu_file(a, n)
    try:
        if write is not None:
            with open(f, "w") as fh:
                json.dump(write, fh)
            return write
        with open(f) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return None


def spread(x):
    """p50 / min / max of a per-step series (HIP events per step, bh_step_times)."""
    x = np.asarray(x, dtype=np.float64)
    if len(x) == 0:
        return None
    return {"p50": float(np.median(x)), "min": float(x.min()), "max": float(x.max()), "steps": int(len(x))}


def step_spread(e):
    st, wk = e.step_times()
    return {"step_ms": spread(st), "walk_ms": spread(wk)}


def timed_leg(G, cfg, mass, pos, vel, steps, warmup, want_state=True):
    """One single-GPU leg: warm-up, K steps enqueued back to back, wall time around them."""
    with G.BarnesHutEngine(cfg) as e:
        e.upload(pos, vel, mass)
        e.step(warmup)
        e.sync()
        t0 = time.perf_counter()
        e.step(steps)
        e.sync()
        dt = time.perf_counter() - t0
        st = e.stats()
        sp = step_spread(e)
        p0 = p1 = None
        if want_state:
            p0, _ = e.download()
            e.step(1)
            p1, _ = e.download()
    n = len(mass)
    return {"value": n * steps / dt, "unit": "body-steps/s", "ms_per_step": dt / steps * 1e3, "build_ms": st.build_ms,
            "walk_ms": st.walk_ms, "keys_ms": st.keys_ms, "sort_ms": st.sort_ms, "scan_ms": st.scan_ms,
            "nodes_ms": st.nodes_ms, "build_bytes_per_body": st.build_bytes / max(n, 1), "per_step": sp}, p0, p1


def committed_traffic(tag):
    """HBM bytes per walk launch from the committed rocprofv3 PMC passes of the same command (counters cannot be read from
    inside the process): profiles/latest_walk_traffic[_<tag>].json, written by scripts/summarize_profile.py, which stamps
    the digest of the device sources -- a profile of OTHER kernels than the ones running is not reported."""
    name = "latest_walk_traffic.json" if tag == "C3" else f"latest_walk_traffic_{tag}.json"
    digest = _digest()
    try:
        with open(os.path.join(ROOT, "profiles", name)) as fh:
            tj = json.load(fh)
    except (OSError, ValueError):
        return {"traffic": None, "traffic_source": None, "traffic_source_digest": None, "source_digest": digest}
    ok = tj.get("source_digest") == digest
    return {"traffic": tj.get("traffic_bytes") if ok else None, "traffic_source": tj.get("source") if ok else None,
            "traffic_source_digest": tj.get("source_digest"), "source_digest": digest}


def step_roofline(ms_per_step, n, u64, build_bytes_per_body=None, kind="f32"):
    """The WHOLE step against the HBM peak: algorithmic bytes per body-step x N / time per step.
    fp32 state: SURVEY's figure, ~270 B of fixed pipeline + 12 + 8 + 20 U64 of walk with this leg's measured U64
    (0.56 KB at C2, 0.60 KB at C3).  fp64 legs: the engine's own per-kernel count for the build (bh_stats_t.build_bytes)
    + 92 + 40 U64."""
    if kind == "f32":
        per_body = SURVEY_PIPELINE_BYTES + SURVEY_WALK_FIXED_BYTES + NODE_BYTES * u64
    else:
        per_body = (build_bytes_per_body or 0.0) + F64_BODY_BYTES + F64_NODE_BYTES * u64
    ach = per_body * n / (ms_per_step * 1e-3) / 1e9
    return {"bound": "hbm", "bytes_per_body_step": per_body, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS}


def walk_roofline(G, cfg, mass, pos, vel, walk_ms, stats_flag, kind="f32"):
    """Algorithmic bytes of ONE walk + integrate launch on this state (counting variant of the kernel, untimed)
    over the measured kernel time.  kind f32: 44 B per body + 20 B per node a wavefront evaluates;
    f64 / exact (the fp64 modes: 32-byte node + 8-byte link): 92 B per body + 40 B per node."""
    import dataclasses
    c2 = dataclasses.replace(cfg, flags=cfg.flags | stats_flag)
    with G.BarnesHutEngine(c2) as se:
        se.upload(pos, vel, mass)
        se.compute_forces()
        ss = se.stats()
    n = len(mass)
    body_b, node_b = (44, NODE_BYTES) if kind == "f32" else (F64_BODY_BYTES, F64_NODE_BYTES)
    b = n * body_b + ss.wave_nodes * node_b
    ach = b / (walk_ms * 1e-3) / 1e9
    r = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
         "kernel_ms": walk_ms, "algorithmic_bytes_per_launch": b, "u64_nodes_per_body": ss.wave_nodes / n,
         "interactions_per_body": ss.interactions / n, "n_nodes": ss.n_nodes}
    simd_cycles = N_SIMDS * walk_ms * 1e-3 * SHADER_CLOCK_HZ
    if kind == "f32" and ss.wave_quads:
        valu = ss.wave_nodes * VALU_CYCLES_PER_CHILD + ss.wave_quads * VALU_CYCLES_PER_QUAD
        r["issue_frac"] = valu / simd_cycles
    elif kind == "f64" and ss.wave_quads:
        valu = (ss.wave_nodes * F64_CYCLES_PER_NODE + ss.wave_accepts * F64_CYCLES_PER_ACCEPTED
                + ss.wave_quads * F64_CYCLES_PER_QUAD)
        r["issue_frac"] = valu / simd_cycles
        r["issue"] = {"valu_cycles_per_launch": valu, "simd_cycles_per_launch": simd_cycles, "nodes_per_launch": ss.wave_nodes,
                      "accepted_nodes_per_launch": ss.wave_accepts, "quads_per_launch": ss.wave_quads,
                      "cycles_per_node": F64_CYCLES_PER_NODE, "cycles_per_accepted_node": F64_CYCLES_PER_ACCEPTED,
                      "cycles_per_quad": F64_CYCLES_PER_QUAD, "clock_hz_assumed": SHADER_CLOCK_HZ}
    elif kind == "exact":
        valu = ss.wave_nodes * EXACT_CYCLES_PER_NODE + ss.wave_accepts * EXACT_CYCLES_PER_ACCEPTED
        r["issue_frac"] = valu / simd_cycles
        r["issue"] = {"valu_cycles_per_launch": valu, "simd_cycles_per_launch": simd_cycles, "nodes_per_launch": ss.wave_nodes,
                      "accepted_nodes_per_launch": ss.wave_accepts, "cycles_per_node": EXACT_CYCLES_PER_NODE,
                      "cycles_per_accepted_node": EXACT_CYCLES_PER_ACCEPTED, "clock_hz_assumed": SHADER_CLOCK_HZ}
    return r


def other_configs(G, IC, local, seed, stats_flag, which):
    """The other BASELINE configurations that fit one GPU: not the headline, the same kernels."""
    out = {}
    table = {
        # tag: (kind, n, theta, precision, steps, warmup)
        "C2": ("uniform", 65536, 0.5, G.Precision.F32, 1000, 20),
        "C4": ("plummer", 1 << 22, 0.5, G.Precision.F32, 10, 2),
        "C5": ("plummer", 1 << 24, 0.3, G.Precision.MIXED, 5, 1),
    }
    for tag in which:
        kind, n, theta, prec, steps, warm = table[tag]
        m, p, v = IC.make(kind, n, seed, quasi_static=True)
        cfg = G.BhConfig(capacity=n, theta=theta, max_depth=21, precision=prec, reference_compat=False, device=local)
        leg, _, _ = timed_leg(G, cfg, m, p, v, steps, warm, want_state=False)
        leg["roofline"] = walk_roofline(G, cfg, m, p, v, leg["walk_ms"], stats_flag)
        leg["roofline"].update(committed_traffic(tag))
        leg["step_roofline"] = step_roofline(leg["ms_per_step"], n, leg["roofline"]["u64_nodes_per_body"])
        leg["minteractions_per_s"] = leg["roofline"]["interactions_per_body"] * leg["value"] / 1e6
        out[tag] = {"workload": f"{kind}_N{n}_theta{theta}_{'mixed' if prec == G.Precision.MIXED else 'f32'}", "steps": steps,
                    "warmup": warm, **leg}
        del m, p, v
    return out


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(spawn_ranks(a, sys.argv[1:]))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: launch it as\n"
                         f"  python -m torch.distributed.run --nnodes=1 --nproc-per-node {a.gpus} --master-addr 127.0.0.1 "
                         f"--master-port P bench.py --gpus {a.gpus} ...\n(or plainly `python bench.py --gpus {a.gpus}`, "
                         f"which starts the ranks itself)")
    import torch
    import torch.distributed as dist
    import gpu_nbody_simulation_amd as G
    from gpu_nbody_simulation_amd import initial_conditions as IC
    from gpu_nbody_simulation_amd.distributed import (LetStepper, ShardedStepper, init_process_group_from_env,
                                                      partition_orb)
    from gpu_nbody_simulation_amd.engine import FLAG_LDS_STACK, FLAG_WALK_STATS

    rank, local, world = init_process_group_from_env("gloo" if a.dry_run else a.backend)
    if a.dry_run:
        t = torch.tensor([rank], dtype=torch.int64)
        if world > 1:
            dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "sum_of_ranks": int(t.item()), "gpus_arg": a.gpus,
                              "steps": a.steps, "warmup": a.warmup, "decomposition": a.decomposition,
                              "launched_by": "torch.distributed.run" if "RANK" in os.environ else "direct"}))
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return
    if a.force_sharded and world == 1 and not dist.is_initialized() and "RANK" in os.environ:
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the process group has {world} rank(s)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    if "BHGPU_REHEARSE_ON_DEVICE" in os.environ:
        # rehearsal aid: several ranks on ONE device (only where the collective library allows it)
        local = int(os.environ["BHGPU_REHEARSE_ON_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    n = a.n_bodies
    prec = G.Precision.MIXED if a.precision == "mixed" else G.Precision.F32
    if prec == G.Precision.MIXED and a.decomposition == "replicated" and (a.gpus > 1 or a.force_sharded):
        raise SystemExit("bench.py: --precision mixed runs on one GPU or with --decomposition let")
    flags = FLAG_LDS_STACK if a.lds_stack else 0
    sharded = world > 1 or a.force_sharded
    use_let = sharded and a.decomposition == "let"
    let_info = None
    if use_let:
        # LET decomposition: NO rank ever generates or holds all bodies.  Every rank draws its own share of the
        # chunk-wise defined state (IC.make_share), then the bodies are dealt into ORB domains on the devices
        # (LetStepper.rebalance: distributed histograms, one all_to_all of records) -- twice: by count before
        # the first walk, by the measured per-group cost after a few steps.
        lo_i, hi_i = n * rank // world, n * (rank + 1) // world
        mass, pos, vel = IC.make_share(a.init, n, a.seed, lo_i, hi_i, quasi_static=True)
        cfg = G.BhConfig(capacity=int(1.5 * n / world) + 4096, theta=a.theta, max_depth=a.max_depth,
                         precision=prec, reference_compat=False, device=local, flags=flags)
        eng = G.BarnesHutEngine(cfg)
        eng.upload(pos, vel, mass)                    # this rank's share, resident in HBM from here on
        overlap = a.let_overlap == "on" or (a.let_overlap == "auto" and world >= 4)
        stepper = LetStepper(eng, rank, world, let_cap=1 << 14, device=dev, overlap=overlap,
                             ids=np.arange(lo_i, hi_i, dtype=np.int64))
        stepper.rebalance()
        for _ in range(3):
            stepper.step()
        stepper.rebalance()                           # cost-weighted; also re-sizes the LET blocks (autotune)
        cap = stepper.let_cap
        bodies_here = eng.n
    else:
        mass, pos, vel = IC.make(a.init, n, a.seed, quasi_static=True)
        cfg = G.BhConfig(capacity=n, theta=a.theta, max_depth=a.max_depth, precision=prec,
                         reference_compat=False, device=local, flags=flags)
        eng = G.BarnesHutEngine(cfg)
        if sharded:
            eng.set_stream(torch.cuda.current_stream().cuda_stream)
        eng.upload(pos, vel, mass)                    # bodies resident in HBM from here on
        stepper = ShardedStepper(eng, rank, world, n, dev, force_exchange=a.force_sharded)

    def sync_all():
        """Wait for this rank's stream, then for everybody -- and FAIL TOGETHER: a rank whose tree or LET
        outgrew its capacity raises from eng.sync(); were it to leave alone, the others would wait in the
        next collective for ever.  The all_reduce of the failure flag is the barrier."""
        err = None
        try:
            eng.sync()
        except Exception as ex:                       # noqa: BLE001 -- re-raised below, on every rank
            err = ex
        torch.cuda.synchronize()
        if world > 1:
            flag = torch.tensor([1 if err else 0], dtype=torch.int32, device=dev if a.backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.item()) and err is None:
                err = RuntimeError("another rank failed (its message is on its own stderr)")
        if err is not None:
            raise err

    for _ in range(a.warmup):
        stepper.step()
    sync_all()
    if use_let:
        stepper.profile = True                         # events at the phase boundaries of every timed step
    t0 = time.perf_counter()
    if not sharded:
        eng.step(a.steps)                            # K steps enqueued back to back on one stream
    else:
        for _ in range(a.steps):
            stepper.step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    phases = None
    if use_let:
        stepper.profile = False
        ph = stepper.phase_ms()
        lst = eng.stats()
        ph["let_tree_last_step"], ph["let_pack_last_step"] = lst.let_tree_ms, lst.let_pack_ms
        keys = sorted(k for k in ph if k != "steps_profiled")
        t = torch.tensor([ph[k] for k in keys], dtype=torch.float64)
        tmax, tsum = t.clone(), t.clone()
        if world > 1:
            tmax = tmax.to(dev) if a.backend == "nccl" else tmax
            tsum = tsum.to(dev) if a.backend == "nccl" else tsum
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
            tmax, tsum = tmax.cpu(), tsum.cpu()
        phases = {"unit": "ms per step, HIP events on each rank's stream, reduced over ranks",
                  "steps_profiled": ph["steps_profiled"],
                  "max_over_ranks": {k: float(x) for k, x in zip(keys, tmax)},
                  "mean_over_ranks": {k: float(x) / world for k, x in zip(keys, tsum)}}
        largest = stepper.check()                    # raises if a LET outgrew its block: run invalid
        cnt = torch.zeros(world, dtype=torch.int64)
        cnt[rank] = int(eng.n)
        if world > 1:
            cnt = cnt.to(dev) if a.backend == "nccl" else cnt
            dist.all_reduce(cnt)
            cnt = cnt.cpu()
        let_info = {"let_cap_quads": cap, "largest_let_quads": largest,
                    "all_to_all_bytes_per_rank_per_step": cap * 80 * (world - 1),
                    "bodies_on_rank0": int(bodies_here), "bodies_per_rank_balanced": n / world,
                    # what a scaling record needs to explain itself: the shares at the end of the run, and the one-GPU
                    # time of THIS library on THIS workload if a `--gpus 1` run left it on this machine (same digest)
                    "efficiency_inputs": {"bodies_per_rank_min": int(cnt.min()), "bodies_per_rank_max": int(cnt.max()),
                                          "single_gpu": single_gpu_record(a, n)}}

    st = eng.stats()                                 # HIP events recorded inside the timed region
    walk_ms = st.walk_ms if not sharded else None

    # ---- untimed: counters of one walk on the final state (second engine, stats variant) --------
    ss = None
    if use_let:
        # distributed, like the run itself: every rank walks ITS final bodies once more with the counting
        # kernels (own tree + LETs of the others); the counters and Newton's third law are all-reduced
        pf, vf = eng.download()
        mf, idf = eng.masses(), eng.ids()
        se = G.BarnesHutEngine(G.BhConfig(capacity=max(len(mf), 1), theta=a.theta, max_depth=a.max_depth, precision=prec,
                                          reference_compat=False, device=local, flags=flags | FLAG_WALK_STATS))
        se.upload(pf, vf, mf)
        sst = LetStepper(se, rank, world, let_cap=cap, device=dev, ids=idf)
        sst.step(integrate=False)
        sst.check()
        s1 = se.stats()
        acc = se.accelerations()
        red = torch.tensor([float(s1.interactions), float(s1.wave_nodes), float(s1.n_nodes),
                            *(mf[:, None] * acc).sum(0), (mf[:, None] * np.abs(acc)).sum()], dtype=torch.float64)
        if world > 1:
            red = red.to(dev) if a.backend == "nccl" else red
            dist.all_reduce(red)
            red = red.cpu()
        let_info["net_force_over_sum_abs_force"] = float(max(abs(red[3]), abs(red[4])) / max(float(red[5]), 1e-300))
        # forces inside each rank's own tree cancel whatever the exchange did, so Newton's third law does
        # not notice a stale or mis-routed LET.  A DISTRIBUTED DIRECT SUM does: every rank names 64 of its bodies,
        # the sample positions are all-gathered (W x 64 x 2 doubles), every rank sums the pull of ITS bodies on ALL
        # samples in fp64 (main_approach_1.cpp:53-75), one all_reduce(SUM) completes the sums, and each rank
        # compares its own samples with what the forest walk gave them.  A missing or wrong remote tree is off by O(1).
        ns = 64
        pick = np.linspace(0, max(len(mf) - 1, 0), ns).astype(np.int64) if len(mf) else np.zeros(ns, dtype=np.int64)
        mine = torch.tensor(pf[pick] if len(mf) else np.zeros((ns, 2)), dtype=torch.float64)
        allp = torch.zeros((world, ns, 2), dtype=torch.float64)
        if world > 1:
            src = mine.to(dev) if a.backend == "nccl" else mine
            dst = allp.to(dev) if a.backend == "nccl" else allp
            dist.all_gather_into_tensor(dst.view(-1), src.view(-1))
            allp = dst.cpu()
        else:
            allp[0] = mine
        sp = allp.view(-1, 2).numpy()
        part = np.zeros_like(sp)
        for c0 in range(0, len(sp), 16):
            d = pf[None, :, :] - sp[c0:c0 + 16, None, :]
            r2 = (d * d).sum(2)
            r2[r2 == 0] = np.inf                                          # the sample itself
            part[c0:c0 + 16] = (mf[None, :, None] * d / (r2 * np.sqrt(r2))[:, :, None]).sum(1)
        tot = torch.tensor(part * cfg.G, dtype=torch.float64)
        if world > 1:
            tot = tot.to(dev) if a.backend == "nccl" else tot
            dist.all_reduce(tot)
            tot = tot.cpu()
        ref = tot.view(world, ns, 2)[rank].numpy()
        rerr = np.linalg.norm(acc[pick] - ref, axis=1) / np.linalg.norm(ref, axis=1) if len(mf) else np.zeros(ns)
        chk = torch.tensor([float(np.median(rerr)), float(np.quantile(rerr, 0.9))], dtype=torch.float64)
        if world > 1:
            chk = chk.to(dev) if a.backend == "nccl" else chk
            dist.all_reduce(chk, op=dist.ReduceOp.MAX)
            chk = chk.cpu()
        # (the reference's own algorithm -- monopoles, theta 0.5 -- is within 1.3e-2 (median) / 4e-2 (95 %) of the
        # direct sum, measured with the oracle; single bodies whose pulls cancel are off by more than their force)
        let_info["direct_sum_check"] = {"samples_per_rank": ns, "worst_rank_median_rel_err": float(chk[0]),
                                        "worst_rank_p90_rel_err": float(chk[1])}
        se.close()

        class _S:                                    # the fields the report below reads
            interactions, wave_nodes, n_nodes = int(red[0]), int(red[1]), int(red[2])
        ss = _S
    else:
        pf, vf = eng.download()
    out = None
    if rank == 0:
        if ss is None:
            with G.BarnesHutEngine(G.BhConfig(capacity=n, theta=a.theta, max_depth=a.max_depth,
                                              precision=prec, reference_compat=False, device=local,
                                              flags=flags | FLAG_WALK_STATS)) as se:
                se.upload(pf, vf, mass)
                se.compute_forces()
                ss = se.stats()
        u64 = ss.wave_nodes / n                      # distinct nodes per body per 64-body group
        # algorithmic bytes of ONE walk+integrate launch (DESIGN.md "Roofline"):
        #   per body: sorted pos 8 + perm 4 + vel r/w 16 + pos w 8 + accel w 8 = 44 B
        #   per wave-node visit: one 20-byte node (a quarter of an 80-byte quad record)
        walk_bytes = n * 44 + ss.wave_nodes * NODE_BYTES
        value = n * a.steps / elapsed
        roof = None
        if walk_ms:
            ach = walk_bytes / (walk_ms * 1e-3) / 1e9
            # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command: only a profile of
            # THESE kernels on THIS workload counts
            ct = committed_traffic("C3")
            if not (n == 1 << 20 and a.init == "plummer"):
                ct.update(traffic=None, traffic_source=None)
            traffic, tsrc, tdig, digest = ct["traffic"], ct["traffic_source"], ct["traffic_source_digest"], ct["source_digest"]
            roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
                    "traffic_source_digest": tdig, "source_digest": digest,
                    "kernel": "walk_fast_kernel", "kernel_ms": walk_ms,
                    "algorithmic_bytes_per_launch": walk_bytes, "u64_nodes_per_body": u64}
            # the fraction of the SIMDs' vector-issue cycles the kernel's instruction mix needs (the walk is
            # an irregular gather: it is bounded by memory LATENCY and vector issue, not by bytes)
            wq = getattr(ss, "wave_quads", 0)
            if wq:
                valu = ss.wave_nodes * VALU_CYCLES_PER_CHILD + wq * VALU_CYCLES_PER_QUAD
                roof["issue_frac"] = valu / (N_SIMDS * walk_ms * 1e-3 * SHADER_CLOCK_HZ)
                # the in-kernel clock of this kernel, measured on the experiments build of the SAME sources (scripts/
                # walk_timeline.py -> profiles/latest_walk_clock.json; MI355X_MICROARCH.md DVFS item 6): not reported for other kernels
                clk = None
                try:
                    with open(os.path.join(ROOT, "profiles", "latest_walk_clock.json")) as fh:
                        cj = json.load(fh)
                    if cj.get("source_digest") == digest and n == 1 << 20 and a.init == "plummer":
                        clk = cj["clock_hz_median"]
                except (OSError, KeyError, ValueError):
                    pass
                if clk:
                    roof["issue_frac_at_measured_clock"] = valu / (N_SIMDS * walk_ms * 1e-3 * clk)
                roof["issue"] = {"clock_hz_measured": clk,
                                 "valu_cycles_per_launch": valu, "simd_cycles_per_launch": N_SIMDS * walk_ms * 1e-3 * SHADER_CLOCK_HZ,
                                 "quads_per_launch": wq, "children_per_launch": ss.wave_nodes,
                                 "cycles_per_child": VALU_CYCLES_PER_CHILD, "cycles_per_quad": VALU_CYCLES_PER_QUAD,
                                 "clock_hz_assumed": SHADER_CLOCK_HZ}
        out = {
            "metric": "body-steps/sec", "value": value, "unit": "body-steps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if a.precision == "f32" else "f64 state / f32 forces", "data": "synthetic",
            "config": {"workload": f"{a.init}_N{n}_theta{a.theta}", "n_bodies": n, "theta": a.theta,
                       "max_depth": a.max_depth, "init": a.init, "seed": a.seed,
                       "parallelism": "1 GPU" if not sharded else
                       (f"orb x{world} (cuts from distributed histograms, bodies dealt on the devices), local trees + LET all_to_all/step" + (" (overlapped)" if overlap else "") if use_let
                        else f"replicated build, hilbert-range walk x{world}, all_gather/step")},
            "minteractions_per_s": ss.interactions * a.steps / elapsed / 1e6,
            "interactions_per_body": ss.interactions / n,
            "build_ms": st.build_ms, "walk_ms": st.walk_ms, "n_nodes": ss.n_nodes,
            "build_groups_ms": {"keys": st.keys_ms, "sort": st.sort_ms, "scan": st.scan_ms, "nodes": st.nodes_ms},
            "roofline": roof,
        }
        out["step_roofline"] = step_roofline(out["ms_per_step"], n, u64)
        out["library"] = {"build_info": eng.build_info(), "product": _lib_is_product(), "source_digest": _digest()}
        if not sharded:
            out["per_step"] = step_spread(eng)
            single_gpu_record(a, n, write={"ms_per_step": out["ms_per_step"], "steps": a.steps, "build_ms": st.build_ms,
                                           "walk_ms": st.walk_ms, "source_digest": _digest()})
        out["rccl"] = ({"world_size": dist.get_world_size(), "backend": dist.get_backend()} if dist.is_initialized()
                       else {"world_size": 1, "backend": None})
        if phases:
            out["phases"] = phases
        if let_info:
            out["let"] = let_info
        if world == 1 and not a.no_secondary:
            # the uniform distribution next to the Plummer one (same N, theta, steps)
            other = "uniform" if a.init == "plummer" else "plummer"
            m2, p2, v2 = IC.make(other, n, a.seed, quasi_static=True)
            leg, _, _ = timed_leg(G, cfg, m2, p2, v2, a.steps, a.warmup)
            leg["roofline"] = walk_roofline(G, cfg, m2, p2, v2, leg["walk_ms"], FLAG_WALK_STATS)
            leg["step_roofline"] = step_roofline(leg["ms_per_step"], n, leg["roofline"]["u64_nodes_per_body"])
            out["secondary"] = {"workload": f"{other}_N{n}_theta{a.theta}", **leg}
            # The headline workload is quasi-static (bodies do not change cells, so the sort permutation of
            # every timed step is the identity).  DYNAMIC leg: same N, theta, dtype and distribution, every
            # body drifting --drift-cells depth-12 cell widths per step in a random direction (masses still
            # tiny: no close-encounter blow-up); reported with the measured churn of the sorted order.
            m3, p3, v3 = IC.make(a.init, n, a.seed, quasi_static=True, drift_cells=a.drift_cells)
            leg, q0, q1 = timed_leg(G, cfg, m3, p3, v3, a.steps, a.warmup)
            leg["roofline"] = walk_roofline(G, cfg, m3, q0, v3, leg["walk_ms"], FLAG_WALK_STATS)
            leg["step_roofline"] = step_roofline(leg["ms_per_step"], n, leg["roofline"]["u64_nodes_per_body"])
            out["dynamic"] = {"workload": f"{a.init}_N{n}_theta{a.theta}_drift{a.drift_cells}cells_per_step",
                              **leg, **rank_churn(q0, q1)}
            # The bit-exact fp64 mode (the parity anchor: the reference's own arithmetic, project.cu:38-65):
            # its throughput at this size (uncapped: max_depth 21) and at BASELINE config 1's size and cap.
            ex = {}
            for tag, (nn, md, kind) in {"C3": (n, a.max_depth, a.init), "C1": (1024, 10, "uniform")}.items():
                me, pe, ve = (mass, pos, vel) if nn == n else IC.make(kind, nn, a.seed, quasi_static=True)
                cfg_e = G.BhConfig(capacity=nn, theta=a.theta, max_depth=md, precision=G.Precision.F64_EXACT,
                                   reference_compat=True, device=local)
                ks = max(3, a.steps // 4) if nn == n else 200
                leg, _, _ = timed_leg(G, cfg_e, me, pe, ve, ks, 2)
                leg["roofline"] = walk_roofline(G, cfg_e, me, pe, ve, leg["walk_ms"], FLAG_WALK_STATS, kind="exact")
                leg["roofline"]["kernel"] = "walk_exact_kernel"
                if nn <= 12288:
                    # (launches this small walk with one wavefront per body, breadth-first -- csrc/bh_walk_exact.hpp,
                    # walk_exact_bfs_kernel; the counters above are the cooperative walk's, the issue model is not this kernel's)
                    leg["roofline"]["kernel"] = "walk_exact_bfs_kernel"
                    leg["roofline"]["issue_frac"] = None
                    leg["roofline"].pop("issue", None)
                if tag == "C3" and nn == 1 << 20 and kind == "plummer":
                    leg["roofline"].update(committed_traffic("EXACT"))       # (profiles/r04_exact: this workload)
                leg["step_roofline"] = step_roofline(leg["ms_per_step"], nn, leg["roofline"]["u64_nodes_per_body"],
                                                     leg["build_bytes_per_body"], kind="f64")
                ex[tag] = {"workload": f"{kind}_N{nn}_theta{a.theta}_depth{md}_exact_fp64", "steps": ks, **leg}
            out["secondary_exact"] = ex
            # BH_PRECISION_F64: the same fp64 tree, the throughput walk (hand-written loop, free order, four siblings per
            # scalar load, criterion on d2, v_rsq_f64 + one Newton step): the reference's arithmetic TYPE at speed;
            # <= 1e-12 of the oracle, same per-body counts
            cfg_f = G.BhConfig(capacity=n, theta=a.theta, max_depth=a.max_depth, precision=G.Precision.F64,
                               reference_compat=True, device=local)
            leg, _, _ = timed_leg(G, cfg_f, mass, pos, vel, max(5, a.steps // 2), 2, want_state=False)
            leg["roofline"] = walk_roofline(G, cfg_f, mass, pos, vel, leg["walk_ms"], FLAG_WALK_STATS, kind="f64")
            leg["roofline"]["kernel"] = "walk_f64_kernel"
            leg["roofline"].update(committed_traffic("F64"))
            leg["step_roofline"] = step_roofline(leg["ms_per_step"], n, leg["roofline"]["u64_nodes_per_body"],
                                                 leg["build_bytes_per_body"], kind="f64")
            out["secondary_f64"] = {"workload": f"{a.init}_N{n}_theta{a.theta}_depth{a.max_depth}_fast_fp64", **leg}
            # ... and at BASELINE config 1's size and cap (N = 1,024, cap 10), next to secondary_exact.C1: one wave's chain
            m1, p1, v1 = IC.make("uniform", 1024, a.seed, quasi_static=True)
            cfg_1 = G.BhConfig(capacity=1024, theta=a.theta, max_depth=10, precision=G.Precision.F64, reference_compat=True,
                               device=local)
            leg, _, _ = timed_leg(G, cfg_1, m1, p1, v1, 200, 2, want_state=False)
            out["secondary_f64_c1"] = {"workload": f"uniform_N1024_theta{a.theta}_depth10_fast_fp64", "steps": 200, **leg}
        if world == 1 and not a.no_secondary and a.other_configs:
            out["other_configs"] = other_configs(G, IC, local, a.seed, FLAG_WALK_STATS,
                                                 [t for t in a.other_configs.split(",") if t])
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(mass, pos, a.theta, a.cpu_sample)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
    eng.close()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
