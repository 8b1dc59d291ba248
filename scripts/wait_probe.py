"""Where does a wave's time go?  Diagnostic build with s_memtime stamps around the traversal loop's s_waitcnt
(scripts/build_variants.sh probe:"-DBH_X_WAITPROBE=1"):
    BHGPU_LIB_OPT_IN=1 BHGPU_LIB=gpu-nbody-simulation_amd/build/libbhgpu_probe.so python scripts/wait_probe.py
Per workload: cycles per loop iteration a wave spends in the wait (minus the stamp pair's own round trip), the
iterations per wave, the walk's time."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpu_nbody_simulation_amd as G  # noqa: E402
from gpu_nbody_simulation_amd import initial_conditions as IC  # noqa: E402
from gpu_nbody_simulation_amd.engine import FLAG_WALK_NO_SPLIT  # noqa: E402

for kind, n in [("plummer", 1 << 20), ("uniform", 1 << 20), ("plummer", 1 << 22), ("plummer", 1 << 18), ("plummer", 1 << 16)]:
    m, p, v = IC.make(kind, n, 1, quasi_static=True)
    with G.BarnesHutEngine(G.BhConfig(capacity=n, theta=0.5, max_depth=21, precision=G.Precision.F32,
                                      reference_compat=False, flags=FLAG_WALK_NO_SPLIT)) as e:
        e.upload(p, v, m)
        e.step(5)
        e.sync()
        e.step(10)
        e.sync()
        st = e.stats()
    waves = max(st.wave_quads, 1)
    iters = max(st.interactions, 1)
    stamp = st.wave_nodes / waves
    print(json.dumps({"kind": kind, "n": n, "walk_ms": round(st.walk_ms, 4), "waves": waves,
                      "iterations_per_wave": round(iters / waves, 1),
                      "stamp_pair_round_trip_cycles": round(stamp, 1),
                      "wait_cycles_per_iteration_raw": round(st.visits / iters, 1),
                      "wait_cycles_per_iteration_minus_stamp": round(st.visits / iters - stamp, 1),
                      "wave_cycles_total_at_2p4GHz_per_iteration": round(st.walk_ms * 1e-3 * 2.4e9 * min(8192, waves) / iters, 1)}), flush=True)
