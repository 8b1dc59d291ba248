"""A workload where the force MOVES the bodies within ten steps and no close encounter blows up: 65,536 bodies
on a 256 x 256 jittered grid over [-0.1, 0.1]^2 (jitter +-0.3 spacings: the closest pair is >= 0.4 spacings = 3.1e-4 apart, far
above the depth-21 leaf size of 2.3e-7), velocities U(-1e-5, 1e-5)^2 -- ten steps move a body by <= 1.4e-4, less than half the
closest pair's distance, so no pair meets (with 1e-4 a few pairs did, and their kicks amplified 1e-7 differences to O(1):
measured) --, equal masses 2.3e-5: the mean-field acceleration is G M / R^2 ~ 1e-8 per step, a neighbour's kick <= G m / d^2 =
1.6e-8, so over 10 steps (dt = 1) the velocities change by ~1e-7 = 1e-2 |v|.  Every other multi-step fp32 workload of the suite
is ballistic by construction (masses 1e-14).
Shared by tests/test_gpu_moving.py and scripts/trajectory_measure.py."""
import numpy as np

from oracle import bh_oracle as O
import gpu_nbody_simulation_amd as G

N_SIDE, STEPS, THETA = 256, 10, 0.5


def make(seed, f32=True):
    r = np.random.default_rng(seed)
    h = 0.2 / N_SIDE
    gx, gy = np.meshgrid(np.arange(N_SIDE), np.arange(N_SIDE), indexing="ij")
    p = np.stack([gx.ravel(), gy.ravel()], axis=1) * h - 0.1 + 0.5 * h + r.uniform(-0.3 * h, 0.3 * h, (N_SIDE * N_SIDE, 2))
    v = r.uniform(-1e-5, 1e-5, p.shape)
    m = np.full(len(p), 2.3e-5)
    order = r.permutation(len(p))                                  # (caller order is not grid order)
    p, v = p[order], v[order]
    if f32:
        p, v, m = (x.astype(np.float32).astype(np.float64) for x in (p, v, m))
    return m, p, v


def oracle_states(m, p, v, steps=STEPS):
    """[(p_k, v_k)] for k = 0 .. steps: the reference's step in fp64 on the uncapped tree."""
    out = [(p.copy(), v.copy())]
    for _ in range(steps):
        pn, vn = O.run(out[-1][0], out[-1][1], m, 1, max_depth=0, theta=THETA)
        out.append((pn, vn))
    return out


def _q(x):
    x = np.asarray(x)
    return {"q50": float(np.median(x)), "q999": float(np.quantile(x, 0.999)), "max": float(x.max())}


def measure(precision, seed):
    """Errors of the velocity CHANGE -- per body relative to its own |dv| (quantiles; a body whose pulls cancel has a small
    |dv| and a large relative error: the maximum is reported relative to the MEDIAN |dv| instead) -- and of the positions in
    units of the box width, teacher-forced (per step, worst step) and free-running (after STEPS steps)."""
    m, p, v = make(seed, f32=(precision == G.Precision.F32))
    ref = oracle_states(m, p, v)
    box = float(np.ptp(p, axis=0).max())
    res = {"dv_over_v_after_10_steps": float(np.median(np.linalg.norm(ref[-1][1] - v, axis=1) / np.linalg.norm(v, axis=1)))}
    cfg = G.BhConfig(capacity=len(m), theta=THETA, max_depth=21, precision=precision, reference_compat=False)
    tf_dv, tf_p, tf_s = [], [], []
    with G.BarnesHutEngine(cfg) as e:
        for k in range(STEPS):
            e.upload(ref[k][0], ref[k][1], m)
            e.step(1)
            pg, vg = e.download()
            dv_ref = ref[k + 1][1] - ref[k][1]
            err = np.linalg.norm((vg - ref[k][1]) - dv_ref, axis=1)
            tf_dv.append(err / np.linalg.norm(dv_ref, axis=1))
            tf_s.append(err / np.median(np.linalg.norm(dv_ref, axis=1)))
            tf_p.append(np.abs(pg - ref[k + 1][0]).max(axis=1) / box)
    res["teacher_forced_dv_rel"] = _q(np.max(tf_dv, axis=0))
    res["teacher_forced_dv_over_median_dv_max"] = float(np.max(tf_s))
    res["teacher_forced_pos_over_box"] = _q(np.max(tf_p, axis=0))
    with G.BarnesHutEngine(cfg) as e:
        e.upload(p, v, m)
        e.step(STEPS)
        pg, vg = e.download()
    dv_ref = ref[-1][1] - v
    err = np.linalg.norm((vg - v) - dv_ref, axis=1)
    res["free_dv_rel"] = _q(err / np.linalg.norm(dv_ref, axis=1))
    res["free_dv_over_median_dv_max"] = float(err.max() / np.median(np.linalg.norm(dv_ref, axis=1)))
    res["free_pos_over_box"] = _q(np.abs(pg - ref[-1][0]).max(axis=1) / box)
    res["min_pair_distance_over_leaf_size"] = 0.4 * (0.2 / N_SIDE) / (1.2 * box / 2 ** 20)
    return res
