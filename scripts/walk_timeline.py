"""Per-wave timeline of the last walk launch (experiments build, BH_WALK_TIMELINE): how full are the wave
slots over the launch, how long is the tail, how do wave durations spread?
And the IN-KERNEL CLOCK of the walk (VERDICT r3 #2a; MI355X_MICROARCH.md, DVFS give-back item 6): per wave
delta s_memtime / delta s_memrealtime x 100 MHz, median over the waves of the LAST walk after `steps` back-to-back steps
(>= 2 s of launches).  Writes profiles-ready JSON when --json PATH is given.
  BHGPU_LIB_OPT_IN=1 BHGPU_LIB=.../libbhgpu_exp.so BH_WALK_TIMELINE=/tmp/tl.bin python scripts/walk_timeline.py [n] [init] [theta] [f32|mixed] [steps] [--json PATH]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpu_nbody_simulation_amd as G
from gpu_nbody_simulation_amd import initial_conditions as IC

import json
argv = [x for x in sys.argv[1:] if not x.startswith("--")]
jpath = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
if jpath: argv.remove(jpath)
n = int(argv[0]) if len(argv) > 0 else 1 << 20
kind = argv[1] if len(argv) > 1 else "plummer"
theta = float(argv[2]) if len(argv) > 2 else 0.5
prec = G.Precision.MIXED if len(argv) > 3 and argv[3] == "mixed" else G.Precision.F32
steps = int(argv[4]) if len(argv) > 4 else 5
path = os.environ["BH_WALK_TIMELINE"]
m, p, v = IC.make(kind, n, 1, quasi_static=True)
with G.BarnesHutEngine(G.BhConfig(capacity=n, theta=theta, max_depth=21, precision=prec, reference_compat=False)) as e:
    e.upload(p, v, m)
    e.step(steps)
    e.sync()
    st = e.stats()
    info = e.build_info()
t = np.fromfile(path, dtype=np.uint64).reshape(-1, 6)
start, end, hw, cost = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64), t[:, 2], t[:, 3].astype(np.int64)
clk = (t[:, 5].astype(np.int64) - t[:, 4].astype(np.int64)) / np.maximum(end - start, 1) * 100e6     # Hz, per wave
long_enough = (end - start) > 2000                                                               # > 20 us: stamp granularity
print(f"in-kernel clock (delta s_memtime / delta s_memrealtime x 100 MHz), waves longer than 20 us: median {np.median(clk[long_enough])/1e9:.4f} GHz  "
      f"p10 {np.percentile(clk[long_enough],10)/1e9:.4f}  p90 {np.percentile(clk[long_enough],90)/1e9:.4f}  ({int(long_enough.sum())} waves, after {steps} steps)")
if jpath:
    from gpu_nbody_simulation_amd.build import source_digest
    json.dump({"kernel": "walk_fast_kernel", "workload": f"{kind}_N{n}_theta{theta}", "clock_hz_median": float(np.median(clk[long_enough])),
               "clock_hz_p10": float(np.percentile(clk[long_enough], 10)), "clock_hz_p90": float(np.percentile(clk[long_enough], 90)),
               "waves": int(long_enough.sum()), "steps_before_the_stamped_launch": steps, "walk_ms_events": st.walk_ms,
               "method": "per wave: delta s_memtime / delta s_memrealtime x 100 MHz; experiments build of the same sources",
               "library": info, "source_digest": source_digest()}, open(jpath, "w"), indent=1)
t0, t1 = start.min(), end.max()
dur = (end - start) * 10.0                      # ns
span = (t1 - t0) * 10.0
print(f"walk_ms (events) {st.walk_ms:.4f}   timeline span {span/1e6:.4f} ms   waves {len(t)}")
print(f"wave duration ns: mean {dur.mean():.0f}  p10 {np.percentile(dur,10):.0f}  p50 {np.percentile(dur,50):.0f}  p90 {np.percentile(dur,90):.0f}  p99 {np.percentile(dur,99):.0f}  max {dur.max():.0f}")
print(f"sum of wave durations / (span x 8192 slots) = {dur.sum() / (span * 8192):.3f}")
print(f"cost (loop iterations): mean {cost.mean():.1f} p10 {np.percentile(cost,10):.0f} p90 {np.percentile(cost,90):.0f} max {cost.max()}  ns per iteration {dur.sum()/cost.sum():.1f}")
# occupancy over time
edges = np.linspace(t0, t1, 21)
for a, b in zip(edges[:-1], edges[1:]):
    live = np.minimum(end, b) - np.maximum(start, a)
    print(f"  t = {(a - t0) * 10 / 1e3:7.1f} us .. {(b - t0) * 10 / 1e3:7.1f} us : {np.clip(live, 0, None).sum() / (b - a):7.0f} waves resident")
late = start > t0 + 0.5 * (t1 - t0)
print(f"waves starting in the second half of the launch: {late.mean():.3f} of all, mean duration {dur[late].mean():.0f} ns vs {dur[~late].mean():.0f} ns")
# per CU balance: hw id bits: wave_id[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13] ... (gfx9 layout)
cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7; xcc = (hw >> 20) & 0xF
key = (xcc.astype(np.int64) << 8) | (se.astype(np.int64) << 4) | cu.astype(np.int64)
busy = np.bincount(np.unique(key, return_inverse=True)[1], weights=dur)
print(f"busy time per CU (sum of its waves' durations): min {busy.min()/1e3:.0f} us  mean {busy.mean()/1e3:.0f} us  max {busy.max()/1e3:.0f} us over {len(busy)} CUs")
