"""Per-wave timeline of the last walk launch (experiments build, BH_WALK_TIMELINE): how full are the wave
slots over the launch, how long is the tail, how do wave durations spread?
  BHGPU_LIB_OPT_IN=1 BHGPU_LIB=.../libbhgpu_exp.so BH_WALK_TIMELINE=/tmp/tl.bin python scripts/walk_timeline.py [n] [init]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpu_nbody_simulation_amd as G
from gpu_nbody_simulation_amd import initial_conditions as IC

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
kind = sys.argv[2] if len(sys.argv) > 2 else "plummer"
path = os.environ["BH_WALK_TIMELINE"]
m, p, v = IC.make(kind, n, 1, quasi_static=True)
with G.BarnesHutEngine(G.BhConfig(capacity=n, theta=0.5, max_depth=21, precision=G.Precision.F32, reference_compat=False)) as e:
    e.upload(p, v, m)
    e.step(5)
    e.sync()
    st = e.stats()
t = np.fromfile(path, dtype=np.uint64).reshape(-1, 4)
start, end, hw, cost = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64), t[:, 2], t[:, 3].astype(np.int64)
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
