"""BH_PRECISION_F64_EXACT at N = 1M Plummer, a few steps (for rocprofv3 / PMC passes)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpu_nbody_simulation_amd as G
from gpu_nbody_simulation_amd import initial_conditions as IC
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
m, p, v = IC.make("plummer", n, 1, quasi_static=True)
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # 8 = BH_FLAG_WALK_PORTABLE: the walk as the reference writes it
with G.BarnesHutEngine(G.BhConfig(capacity=n, theta=0.5, max_depth=21, precision=G.Precision.F64_EXACT, reference_compat=True, flags=flags)) as e:
    e.upload(p, v, m); e.step(steps); e.sync()
    st = e.stats()
    print(json.dumps({"walk_ms": st.walk_ms, "build_ms": st.build_ms}))
