"""fp64 walks, bodies per wavefront (BH_EXACT_BPW: 64 = rounds 1-3, 0 = by launch size): step / walk ms over N for the bit-exact and the
throughput mode.   python scripts/bpw_ab.py [sizes]"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "--worker":
    import numpy as np
    import gpu_nbody_simulation_amd as G
    from gpu_nbody_simulation_amd import initial_conditions as IC
    n = int(sys.argv[2])
    m, p, v = IC.make("uniform" if n <= 4096 else "plummer", n, 1, quasi_static=True)
    out = {"n": n, "bpw": os.environ.get("BH_EXACT_BPW", "auto")}
    for prec in (G.Precision.F64_EXACT, G.Precision.F64):
        with G.BarnesHutEngine(G.BhConfig(capacity=n, theta=0.5, max_depth=10 if n <= 4096 else 21, precision=prec, reference_compat=True)) as e:
            e.upload(p, v, m); e.step(10); e.sync()
            k = 200 if n <= 65536 else 30
            t0 = time.perf_counter(); e.step(k); e.sync(); dt = (time.perf_counter() - t0) / k * 1e3
            st = e.stats(); pp, vv = e.download()
        out[prec.name] = {"ms_per_step": round(dt, 4), "walk_ms": round(st.walk_ms, 4), "state_hash": hash(pp.tobytes() + vv.tobytes())}
    print(json.dumps(out)); sys.exit(0)
sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1024,4096,16384,65536,131072,262144").split(",")]
for n in sizes:
    for b in (os.environ.get("BPW_LIST", "64,0").split(",")):
        env = dict(os.environ, BH_EXACT_BPW=b)
        subprocess.run([sys.executable, __file__, "--worker", str(n)], env=env)
