"""ms/step and walk ms for every split factor of the fp32 walk (BH_WALK_SPLIT: waves per 64-body
group; 1 = depth-first one-wave walk; 0 = the engine's automatic choice) over N: the data behind the
choice in enqueue_walk (bh_engine.hip).   python scripts/split_sweep.py [sizes] [inits]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpu_nbody_simulation_amd as G
from gpu_nbody_simulation_amd import initial_conditions as IC
sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1024,4096,16384,32768,65536,131072,262144,524288").split(",")]
inits = (sys.argv[2] if len(sys.argv) > 2 else "plummer,uniform").split(",")
rows = []
for init in inits:
    for n in sizes:
        m, p, v = IC.make(init, n, 1, quasi_static=True)
        out = []
        for S in (1, 2, 4, 8, 16, 0):
            os.environ["BH_WALK_SPLIT"] = str(S)
            with G.BarnesHutEngine(G.BhConfig(capacity=n, theta=0.5, max_depth=21, precision=G.Precision.F32, reference_compat=False)) as e:
                e.upload(p, v, m); e.step(5); e.sync()
                t0 = time.perf_counter(); e.step(50); e.sync()
                ms = (time.perf_counter() - t0) / 50 * 1e3
                st = e.stats()
            rows.append({"init": init, "n": n, "split": S, "ms_per_step": ms, "walk_ms": st.walk_ms, "build_ms": st.build_ms})
            out.append("S%d %.3f/%.3f" % (S, ms, st.walk_ms))
        print(init, n, "  ".join(out), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/split_sweep.json", "w"), indent=1)
