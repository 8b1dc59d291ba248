"""A/B of libbhgpu build variants on the GPU box (no torch, one subprocess per library):
bit-identity of the hand-scheduled walk against the C++ loop, then walk / step times of the bench workload.
    python scripts/lib_ab.py [--libs a.so b.so ...] [--n 1048576] [--reps 3] [--steps 30]
Without --libs: the product library and every gpu-nbody-simulation_amd/build/libbhgpu_*.so."""
import argparse
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(a):
    import numpy as np
    import gpu_nbody_simulation_amd as G
    from gpu_nbody_simulation_amd import initial_conditions as IC
    from gpu_nbody_simulation_amd.engine import FLAG_WALK_NO_SPLIT, FLAG_WALK_PORTABLE

    def cached(kind, n):
        f = f"/tmp/ic_{kind}_{n}.npz"
        if not os.path.exists(f):
            m, p, v = IC.make(kind, n, 1, quasi_static=True)
            np.savez(f, m=m, p=p, v=v)
        z = np.load(f)
        return z["m"], z["p"], z["v"]

    out = {"lib": os.path.basename(os.environ.get("BHGPU_LIB", "libbhgpu.so"))}
    # ---- bit-identity: asm loop vs C++ loop (one wave per group), incl. bucket leaves
    ok = True
    if not a.no_check:
        rng = np.random.default_rng(5)
        f32 = lambda x: np.asarray(x, dtype=np.float64).astype(np.float32).astype(np.float64)
        cases = [(*IC.make("plummer", 40000, 3, quasi_static=True), 21, False)]
        n = 30000
        p = f32(np.concatenate([rng.normal(0, 1e-3, (n // 2, 2)), rng.uniform(-1, 1, (n - n // 2, 2))]))
        cases.append((f32(rng.uniform(0.1, 0.5, n)), p, f32(rng.uniform(-1e-9, 1e-9, (n, 2))), 8, False))
        for m, p, v, md, compat in cases:
            res = []
            for flags in (FLAG_WALK_NO_SPLIT, FLAG_WALK_NO_SPLIT | FLAG_WALK_PORTABLE):
                with G.BarnesHutEngine(G.BhConfig(capacity=len(m), max_depth=md, precision=G.Precision.F32,
                                                  reference_compat=compat, flags=flags)) as e:
                    e.upload(p, v, m)
                    e.compute_forces()
                    acc = e.accelerations()
                    e.step(2)
                    res.append((acc,) + e.download())
            ok = ok and all(np.array_equal(x, y) for x, y in zip(*res)) and np.isfinite(res[0][0]).all()
    out["bitwise_equal_portable"] = bool(ok)
    # ---- timing
    for kind, n in [("plummer", a.n)] + ([("uniform", a.n), ("plummer", 131072)] if a.more else []):
        m, p, v = cached(kind, n)
        with G.BarnesHutEngine(G.BhConfig(capacity=n, theta=0.5, max_depth=21, precision=G.Precision.F32,
                                          reference_compat=False)) as e:
            e.upload(p, v, m)
            e.step(5)
            e.sync()
            walks, steps = [], []
            for _ in range(a.reps):
                t0 = time.perf_counter()
                e.step(a.steps)
                e.sync()
                steps.append((time.perf_counter() - t0) / a.steps * 1e3)
                walks.append(e.stats().walk_ms)
            st = e.stats()
        out[f"{kind}_{n}"] = {"walk_ms": [round(x, 4) for x in walks], "step_ms": [round(x, 4) for x in steps],
                              "build_ms": round(st.build_ms, 4)}
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="*")
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--more", action="store_true", help="also uniform at N and Plummer at 131,072")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--worker", action="store_true")
    a = ap.parse_args()
    if a.worker:
        return worker(a)
    pkg = os.path.join(ROOT, "gpu-nbody-simulation_amd")
    libs = a.libs or [os.path.join(pkg, "libbhgpu.so")] + sorted(glob.glob(os.path.join(pkg, "build", "libbhgpu_*.so")))
    for rnd in range(2):                                  # two rounds: box-to-box drift shows as a difference between them
        for lib in libs:
            env = dict(os.environ, BHGPU_LIB=os.path.abspath(lib), BHGPU_LIB_OPT_IN="1")
            cmd = [sys.executable, os.path.abspath(__file__), "--worker", "--n", str(a.n), "--reps", str(a.reps),
                   "--steps", str(a.steps)] + (["--more"] if a.more else []) + (["--no-check"] if (a.no_check or rnd) else [])
            try:
                r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
                line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "rc=%d %s" % (r.returncode, r.stderr[-300:])
            except subprocess.TimeoutExpired:
                line = json.dumps({"lib": os.path.basename(lib), "error": "timeout"})
                print(line, flush=True)
                return 1                                   # a hung kernel: run nothing else on this box
            print(line, flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
