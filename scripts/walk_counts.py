"""Quads evaluated per 64-body wave (the length of a wave's dependent chain) for several N."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpu_nbody_simulation_amd as G
from gpu_nbody_simulation_amd import initial_conditions as IC
from gpu_nbody_simulation_amd.engine import FLAG_WALK_STATS, FLAG_WALK_NO_SPLIT
for init in ("plummer", "uniform"):
    for n in (16384, 65536, 262144, 1048576):
        m, p, v = IC.make(init, n, 1, quasi_static=True)
        with G.BarnesHutEngine(G.BhConfig(capacity=n, theta=0.5, max_depth=21, precision=G.Precision.F32,
                                          reference_compat=False, flags=FLAG_WALK_STATS | FLAG_WALK_NO_SPLIT)) as e:
            e.upload(p, v, m); e.compute_forces(); st = e.stats()
        waves = (n + 63) // 64
        print(f"{init} N={n}: nodes/wave {st.wave_nodes / waves:.0f} = quads/wave {st.wave_nodes / waves / 4:.0f}; "
              f"visits/body {st.visits / n:.0f}; interactions/body {st.interactions / n:.0f}; tree nodes {st.n_nodes}")
