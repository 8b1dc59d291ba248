"""Synthetic initial conditions for sizes beyond the reference's shipped files (40,960 bodies).

uniform(): the distribution of the shipped init files -- positions U(-0.1, 0.1)^2, velocities
U(-1e-4, 1e-4)^2, masses log-uniform 1e-2..1e1 (the files span 0.0100037..9.99933 although
project.cu:30-35 now says 0.1..0.5).  plummer(): BASELINE config 3 -- a 3-D Plummer sphere
(scale a = 0.02, truncated at 10a) projected on (x, y), equal masses, zero velocities.

The reference seeds rand()/cuRAND from time() (project.cu:323, 1051) and is not reproducible;
these use numpy's counter-based Philox so that a (seed, n) pair always gives the same bodies.
Values are rounded to float32 so the fp32 engine and the fp64 oracle see identical inputs.
"""
from __future__ import annotations

import numpy as np


def _rng(seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=seed))


def _f32(a: np.ndarray) -> np.ndarray:
    return a.astype(np.float32).astype(np.float64)


def uniform(n: int, seed: int = 1, total_mass: float | None = None, vel_scale: float = 1e-4):
    """(masses[n], positions[n,2], velocities[n,2]) float64 holding float32-representable values.

    total_mass=None keeps the shipped files' mass scale (mean 1.45 per body).  With G = 6.67e-11
    and dt = 1 that scale has a dynamical time sqrt(R^3/(G*M)) of ~16 steps at N = 40,960 and ~3
    steps at N = 1,048,576: the cloud collapses and then explodes within a handful of steps, after
    which a benchmark no longer measures a uniform distribution.  Benchmarks therefore pass
    total_mass (the log-uniform masses are rescaled to that sum) and a matching vel_scale."""
    r = _rng(seed)
    pos = r.uniform(-0.1, 0.1, size=(n, 2))
    vel = r.uniform(-vel_scale, vel_scale, size=(n, 2))
    mass = 10.0 ** r.uniform(-2.0, 1.0, size=n)
    if total_mass is not None:
        mass *= total_mass / mass.sum()
    return _f32(mass), _f32(pos), _f32(vel)


def plummer(n: int, seed: int = 1, a: float = 0.02, rmax_over_a: float = 10.0, total_mass: float = 1.0):
    """Equal masses total_mass / n, zero velocities."""
    mass = total_mass / n
    r = _rng(seed)
    out = np.empty((n, 2))
    filled = 0
    while filled < n:
        k = int((n - filled) * 1.05) + 16
        u = r.uniform(0.0, 1.0, size=k)
        rad = a / np.sqrt(np.maximum(u, 1e-300) ** (-2.0 / 3.0) - 1.0)
        cz = r.uniform(-1.0, 1.0, size=k)
        phi = r.uniform(0.0, 2.0 * np.pi, size=k)
        ok = rad <= rmax_over_a * a
        s = np.sqrt(1.0 - cz[ok] ** 2) * rad[ok]
        xy = np.stack([s * np.cos(phi[ok]), s * np.sin(phi[ok])], axis=1)
        take = min(len(xy), n - filled)
        out[filled:filled + take] = xy[:take]
        filled += take
    return _f32(np.full(n, mass)), _f32(out), np.zeros((n, 2))


# The reference skips a node whose mass is <= 1e-15 together with its whole subtree (project.cu:617);
# the fp32 node kernel stores such a node as empty.  Benchmark masses must stay above that cutoff.
EMPTY_NODE_MASS = 1e-15
QUASI_STATIC_BODY_MASS = 1e-14     # lightest body of the quasi-static benchmark workloads


def make(kind: str, n: int, seed: int = 1, quasi_static: bool = False, drift_cells: float = 0.0,
         drift_depth: int = 12):
    """quasi_static: the benchmark's mass scale.  The reference has no softening and dt = 1
    (project.cu:29, 633-634).  With unit masses a 10^6-body Plummer sphere has a dynamical time of
    0.35 steps; and whatever the scale, the closest pairs (separations down to ~1e-10 near the
    origin, where fp32 is finest) kick each other by G*m/r^2 per step: measured on MI355X, total
    mass 1e-3 still ejects bodies at 4.7 per step, the root box grows 375x in 23 steps and the
    depth-cap cells turn into buckets -- the benchmark would time that degenerate tree, not the
    stated distribution.  quasi_static scales PER BODY: every mass is >= 1e-14, i.e. above the
    reference's 1e-15 empty-node cutoff (project.cu:617) at every N (a fixed TOTAL mass of 1e-8 put
    the equal Plummer masses below the cutoff from N = 8M up and 39 % of the log-uniform masses at
    N = 1M: those bodies' leaves were stored as empty and their force evaluations skipped).  Plummer:
    equal masses 1e-14; uniform: log-uniform 1e-14..1e-11 (the shipped files' three decades).
    Velocities <= 1e-9, so every timed step sees the distribution as generated.  The work per step
    does not depend on the mass scale.

    drift_cells > 0 (the DYNAMIC benchmark leg): every body additionally moves ballistically by
    drift_cells x (the width of a depth-`drift_depth` cell of the root box) per step in a random
    direction (dt = 1), so a stated fraction of the bodies changes leaf cell -- and sorted rank --
    every step, while the masses stay too small for close-encounter blow-ups."""
    if kind == "uniform":
        if not quasi_static:
            m, p, v = uniform(n, seed)
        else:
            m, p, v = uniform(n, seed, total_mass=None, vel_scale=1e-9)
            m = _f32(m * (QUASI_STATIC_BODY_MASS / 1e-2))       # 1e-2..1e1 -> 1e-14..1e-11
    elif kind == "plummer":
        m, p, v = plummer(n, seed, total_mass=QUASI_STATIC_BODY_MASS * n) if quasi_static else plummer(n, seed)
    else:
        raise ValueError(f"unknown initial condition {kind!r}")
    if quasi_static:
        assert (m.astype(np.float32) > EMPTY_NODE_MASS).all(), "benchmark masses must exceed the empty-node cutoff"
    if drift_cells > 0.0 and n > 0:
        span = float(max(p.max(0) - p.min(0)))
        cell = 1.2 * span / (1 << drift_depth)                  # root box = bounding box + 10 % on every side
        r = _rng(seed + 7919)
        ang = r.uniform(0.0, 2.0 * np.pi, size=n)
        v = _f32(v + drift_cells * cell * np.stack([np.cos(ang), np.sin(ang)], axis=1))
    return m, p, v


SHARE_CHUNK = 65536


def make_share(kind: str, n: int, seed: int, lo: int, hi: int, quasi_static: bool = True):
    """Bodies [lo, hi) of an n-body synthetic state that is defined chunk by chunk (SHARE_CHUNK bodies per
    chunk, chunk c drawn from its own counter-based stream), so that ANY rank can generate exactly its own
    share without generating -- or ever holding -- the rest: the multi-GPU benchmark's input (every rank
    calls this with its own range; the union over ranks is the same state for every world size).  Same
    distributions and mass scale as make(); not the same bodies as make(kind, n, seed), whose Plummer
    rejection sampling consumes its stream sequentially."""
    if not (0 <= lo <= hi <= n):
        raise ValueError("make_share: need 0 <= lo <= hi <= n")
    ms, ps, vs = [], [], []
    for c in range(lo // SHARE_CHUNK, (max(hi, lo + 1) - 1) // SHARE_CHUNK + 1):
        c0 = c * SHARE_CHUNK
        cn = min(SHARE_CHUNK, n - c0)
        if cn <= 0:
            break
        m, p, v = make(kind, cn, seed * 1000003 + c + 1, quasi_static=quasi_static)
        a, b = max(lo, c0) - c0, min(hi, c0 + cn) - c0
        ms.append(m[a:b]); ps.append(p[a:b]); vs.append(v[a:b])
    if not ms:
        return np.zeros(0), np.zeros((0, 2)), np.zeros((0, 2))
    return np.concatenate(ms), np.concatenate(ps), np.concatenate(vs)
