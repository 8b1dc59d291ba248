"""Long-run stability check on the GPU: dynamic (not quasi-static) inputs, thousands of steps, all
precisions; after every leg the state must be finite (fp32 modes) and the tree consistent.
python scripts/stress.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpu_nbody_simulation_amd as G
from gpu_nbody_simulation_amd import initial_conditions as IC

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
ok = True
for prec, name in ((G.Precision.F32, "f32"), (G.Precision.MIXED, "mixed"), (G.Precision.F64_EXACT, "exact")):
    for kind, n, mscale in (("plummer", 65536, 1e-6), ("uniform", 200000, 1e-4), ("plummer", 1 << 20, 1e-8)):
        if prec == G.Precision.F64_EXACT and n > 70000:
            continue
        m, p, v = IC.make(kind, n, 7)
        m = m / m.sum() * mscale * n / 65536                 # dynamic but not instantly explosive
        md = 10 if prec == G.Precision.F64_EXACT else 21
        k = steps if n < (1 << 20) else max(steps // 10, 50)
        t0 = time.perf_counter()
        with G.BarnesHutEngine(G.BhConfig(capacity=n, precision=prec, max_depth=md, reference_compat=(prec == G.Precision.F64_EXACT))) as e:
            e.upload(p, v, m)
            done = 0
            while done < k:
                e.step(min(250, k - done)); done += min(250, k - done)
                pos, vel = e.download()
                if prec != G.Precision.F64_EXACT and not (np.isfinite(pos).all() and np.isfinite(vel).all()):
                    print(f"  NON-FINITE state at step {done}", flush=True); ok = False; break
            st = e.stats()
            e.build_tree()
            nodes, depth = e.export_tree()
            mass_root = nodes[0]["mass"]
        dt = time.perf_counter() - t0
        moved = float(np.median(np.linalg.norm(pos - p, axis=1)))
        good = abs(mass_root - m.sum()) <= 1e-5 * m.sum() + 1e-30 and st.n_bodies == n
        ok &= bool(good)
        print(f"{name:6s} {kind:8s} N={n:8d} steps={k:5d}  {dt:6.1f}s  median displacement {moved:.3e}  nodes {len(nodes)}  "
              f"max depth {int(depth.max())}  root mass ok={good}", flush=True)
print("STRESS", "OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
