"""Developer check: exact + fp32 engine vs the oracle on the golden inputs (prints, no asserts)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import bh_oracle as O
import gpu_nbody_simulation_amd as G
from gpu_nbody_simulation_amd.engine import FLAG_WALK_STATS, FLAG_LDS_STACK

gold = lambda n: np.load(os.path.join(ROOT, "tests", "golden", n + ".npz"))
d = os.path.join(ROOT, "tests", "golden", "init1024")
m = np.loadtxt(d + "/masses_init.txt"); p = np.loadtxt(d + "/positions_init.txt"); v = np.loadtxt(d + "/velocities_init.txt")
g = gold("ref_project_1024")

def canon(nodes):
    c = nodes.copy(); c["child"] = np.where(c["child"] == -1, -1.0, 1.0); return c

print("== exact, N=1024, shipped files")
e = G.BarnesHutEngine(G.BhConfig(capacity=1024, flags=FLAG_WALK_STATS))
e.upload(p, v, m)
e.build_tree()
nodes, depth = e.export_tree()
ref_nodes, ref_depth = O.canonical_tree(g["tree_0"])
print(" n_nodes", len(nodes), "ref", len(ref_nodes))
if len(nodes) == len(ref_nodes):
    print(" depth equal", np.array_equal(depth, ref_depth))
    cn = canon(nodes)
    for f in cn.dtype.names:
        print("  field", f, np.array_equal(cn[f], ref_nodes[f]))
f = e.compute_forces()
print(" forces bitwise", np.array_equal(f, g["forces_0"]), "maxrel", np.abs(f - g["forces_0"]).max() / np.abs(g["forces_0"]).max())
st = e.stats(); print(" stats", st)
e.step(1); pp, vv = e.download()
print(" step0 pos", np.array_equal(pp, g["pos_after_0"]), "vel", np.array_equal(vv, g["vel_after_0"]))
e.upload(p, v, m); e.step(100); pp, vv = e.download()
print(" 100 steps pos", np.array_equal(pp, g["pos_after_99"]), "vel", np.array_equal(vv, g["vel_after_99"]))
e.close()

for name in ("ref_project_4096", "ref_project_4096_grid", "ref_project_40960"):
    gg = gold(name)
    e = G.BarnesHutEngine(G.BhConfig(capacity=len(gg["mass"])))
    e.upload(gg["pos"], gg["vel"], gg["mass"])
    t0 = time.time(); f = e.compute_forces(); t1 = time.time()
    print("==", name, "forces bitwise", np.array_equal(f, gg["forces_0"]), "n_nodes", e.stats().n_nodes, int(gg["tree_0_n_nodes"]), "%.1f ms" % ((t1 - t0) * 1e3))
    if "pos_after_9" in gg:
        e.step(10); pp, vv = e.download()
        print("   10 steps pos", np.array_equal(pp, gg["pos_after_9"]), "vel", np.array_equal(vv, gg["vel_after_9"]))
    e.close()

print("== ma2 uncapped (max_depth=32)")
gg = gold("ref_ma2_1000")
e = G.BarnesHutEngine(G.BhConfig(capacity=1000, max_depth=32))
e.upload(gg["pos"], gg["vel"], gg["mass"])
f = e.compute_forces()
print(" forces bitwise", np.array_equal(f, gg["forces_0"]), np.abs(f - gg["forces_0"]).max())
e.step(10); pp, vv = e.download()
print(" 10 steps", np.array_equal(pp, gg["pos_after_9"]), np.array_equal(vv, gg["vel_after_9"]))
e.close()

print("== fp32 mode")
for name, md in (("ref_project_4096_grid", 10), ("ref_project_40960", 10), ("ref_project_40960", 16)):
    gg = gold(name)
    P = gg["pos"].astype(np.float32).astype(np.float64); M = gg["mass"].astype(np.float32).astype(np.float64)
    V = gg["vel"].astype(np.float32).astype(np.float64)
    t = O.build_tree(P, M, md)
    fo = O.compute_forces(t, P, M)
    ao = fo / M[:, None]
    for flags in (0, FLAG_LDS_STACK):
        e = G.BarnesHutEngine(G.BhConfig(capacity=len(M), max_depth=md, precision=G.Precision.F32, flags=flags | FLAG_WALK_STATS))
        e.upload(P, V, M)
        e.compute_forces(); a = e.accelerations(); st = e.stats()
        rel = np.linalg.norm(a - ao, axis=1) / np.linalg.norm(ao, axis=1)
        print(" ", name, "depth", md, "flags", flags, "n_nodes", st.n_nodes, len(t), "rel err median %.2e p99.9 %.2e max %.2e" % (np.median(rel), np.quantile(rel, 0.999), rel.max()),
              "visits/body %.1f inter/body %.1f" % (st.visits / len(M), st.interactions / len(M)))
        e.close()
print("done")
