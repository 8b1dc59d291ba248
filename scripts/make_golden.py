#!/usr/bin/env python3
"""Generate tests/golden/* from the REFERENCE's own compiled code (development container only).

Needs /root/reference and `make -C oracle ref` (oracle/_ref/ref_* binaries = the reference's
functions + oracle/ref_driver.cpp).  The fixtures are DATA: inputs and the reference's outputs.
No reference source text is stored.

    python scripts/make_golden.py            # regenerates everything under tests/golden/
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF_IMPL = "/root/reference/implementation"
REFBIN = os.path.join(ROOT, "oracle", "_ref")
GOLD = os.path.join(ROOT, "tests", "golden")

from oracle import bh_oracle as O  # noqa: E402  (used only to re-order the reference's tree dump)


def head_lines(src: str, dst: str, n: int) -> None:
    with open(src) as f, open(dst, "w") as g:
        for i, line in enumerate(f):
            if i >= n:
                break
            g.write(line)


def make_init_dir(d: str, n: int) -> None:
    for name in ("masses", "positions", "velocities"):
        head_lines(os.path.join(REF_IMPL, f"{name}_init.txt"), os.path.join(d, f"{name}_init.txt"), n)


def write_init_dir(d: str, mass, pos, vel) -> None:
    with open(os.path.join(d, "masses_init.txt"), "w") as f:
        for m in mass:
            f.write(repr(float(m)) + "\n")
    for name, arr in (("positions", pos), ("velocities", vel)):
        with open(os.path.join(d, f"{name}_init.txt"), "w") as f:
            for x, y in arr:
                f.write(repr(float(x)) + " " + repr(float(y)) + "\n")


def parse_init_dir(d: str, n: int):
    mass = np.loadtxt(os.path.join(d, "masses_init.txt"), max_rows=n)
    pos = np.loadtxt(os.path.join(d, "positions_init.txt"), max_rows=n)
    vel = np.loadtxt(os.path.join(d, "velocities_init.txt"), max_rows=n)
    return mass, pos, vel


def run_ref(binary: str, init_dir: str, n_steps: int, dumps: list[int]) -> str:
    out = tempfile.mkdtemp(prefix="refout_")
    subprocess.check_call([os.path.join(REFBIN, binary), init_dir, str(n_steps), out,
                           ",".join(map(str, dumps))], stdout=subprocess.DEVNULL)
    return out


def rd(out: str, name: str, shape=None, dtype=np.float64):
    a = np.fromfile(os.path.join(out, name), dtype=dtype)
    return a.reshape(shape) if shape else a


def tree_of(out: str, s: int) -> np.ndarray:
    return np.fromfile(os.path.join(out, f"tree_{s}.bin"), dtype=O.NODE_DTYPE)


def tree_digest(nodes: np.ndarray):
    can, depth = O.canonical_tree(nodes)
    h = hashlib.sha256()
    h.update(depth.tobytes())
    h.update(can.tobytes())
    return h.hexdigest(), np.bincount(depth)


def project_case(n: int, init_dir: str, n_steps: int, dumps: list[int], tree_steps: list[int],
                 text_steps: list[int], store_inputs: bool, store_forces_all=False):
    out = run_ref(f"ref_project_{n}", init_dir, n_steps, dumps)
    d: dict[str, np.ndarray] = {}
    if store_inputs:
        m, p, v = parse_init_dir(init_dir, n)
        d["mass"], d["pos"], d["vel"] = m, p, v
    d["dump_steps"] = np.array(dumps)
    for s in dumps:
        d[f"pos_after_{s}"] = rd(out, f"pos_{s}.bin", (n, 2))
        d[f"vel_after_{s}"] = rd(out, f"vel_{s}.bin", (n, 2))
        if s == 0 or store_forces_all:
            d[f"forces_{s}"] = rd(out, f"forces_{s}.bin", (n, 2))
        t = tree_of(out, s)
        dig, hist = tree_digest(t)
        d[f"tree_{s}_n_nodes"] = np.array(len(t))
        d[f"tree_{s}_sha256"] = np.array(dig)
        d[f"tree_{s}_depth_hist"] = hist
        if s in tree_steps:
            d[f"tree_{s}"] = t
        if s in text_steps:
            d[f"quadtree_txt_{s}"] = np.frombuffer(
                open(os.path.join(out, f"quadtree_{s}.txt"), "rb").read(), dtype=np.uint8)
    return d


def main() -> None:
    os.makedirs(GOLD, exist_ok=True)
    if not os.path.isdir(REF_IMPL):
        sys.exit("reference not present; fixtures can only be regenerated in the dev container")

    # -- (1) the reference's own data files, first 1024 lines (BASELINE config[0]) -----------
    init1024 = os.path.join(GOLD, "init1024")
    os.makedirs(init1024, exist_ok=True)
    make_init_dir(init1024, 1024)

    # -- (2) project.cu CPU path, N=1024, 100 steps on the shipped files ----------------------
    d = project_case(1024, init1024, 100, [0, 1, 2, 9, 49, 99], tree_steps=[0, 1],
                     text_steps=[0, 99], store_inputs=False)
    np.savez_compressed(os.path.join(GOLD, "ref_project_1024.npz"), **d)

    # -- (3) N=4096, shipped files, 10 steps ---------------------------------------------------
    with tempfile.TemporaryDirectory() as t:
        make_init_dir(t, 4096)
        d = project_case(4096, t, 10, [0, 1, 4, 9], tree_steps=[0], text_steps=[],
                         store_inputs=True)
    np.savez_compressed(os.path.join(GOLD, "ref_project_4096.npz"), **d)

    # -- (4) N=4096 synthetic jittered grid (no two bodies share a depth-10 cell): the
    #        encounter-free multi-step case.  Values are float32-representable so the fp32
    #        engine sees bit-identical inputs. ---------------------------------------------
    rng = np.random.default_rng(20251004)
    g = 64
    ij = np.stack(np.meshgrid(np.arange(g), np.arange(g), indexing="ij"), -1).reshape(-1, 2)
    h = 0.2 / g
    pos = (-0.1 + (ij + 0.5) * h + rng.uniform(-0.3, 0.3, size=(g * g, 2)) * h)
    perm = rng.permutation(g * g)
    pos = pos[perm].astype(np.float32).astype(np.float64)
    # the shipped files' mass scale moves bodies into shared depth-10 cells (the self-interaction
    # artefact of SURVEY 0 fact 4) by step 2; masses x 1e-4 and velocities <= 4e-6 keep all 20
    # steps encounter-free (asserted below with the oracle on the reference's own trajectory)
    vel = rng.uniform(-4e-6, 4e-6, size=(g * g, 2)).astype(np.float32).astype(np.float64)
    mass = (10.0 ** rng.uniform(-6, -3, size=g * g)).astype(np.float32).astype(np.float64)
    with tempfile.TemporaryDirectory() as t:
        write_init_dir(t, mass, pos, vel)
        d = project_case(4096, t, 20, [0, 1, 2, 4, 9, 19], tree_steps=[0], text_steps=[],
                         store_inputs=True, store_forces_all=True)
    assert np.array_equal(d["pos"], pos) and np.array_equal(d["mass"], mass)
    for s_ in (0, 1, 2, 4, 9, 19):
        t_ = O.build_tree(d[f"pos_after_{s_}"], mass, 10)
        assert not np.any((t_["child"][:, 0] == -1) & (t_["particle"] == -1) & (t_["mass"] > 0)), s_
    np.savez_compressed(os.path.join(GOLD, "ref_project_4096_grid.npz"), **d)

    # -- (5) N=40960 (the reference's published size), step 0 only; tree kept as a digest ----
    with tempfile.TemporaryDirectory() as t:
        make_init_dir(t, 40960)
        d = project_case(40960, t, 1, [0], tree_steps=[], text_steps=[], store_inputs=True)
    d.pop("pos_after_0"); d.pop("vel_after_0")   # derivable from forces_0; keeps the file small
    # inputs carry 6 significant digits: float32 text round-trip is NOT exact, keep float64
    np.savez_compressed(os.path.join(GOLD, "ref_project_40960.npz"), **d)

    # -- (6) main_approach_2.cpp (uncapped tree), N=1000, 10 steps -----------------------------
    with tempfile.TemporaryDirectory() as t:
        make_init_dir(t, 1000)
        out = run_ref("ref_ma2", t, 10, [0, 1, 9])
        m, p, v = parse_init_dir(t, 1000)
    d = {"mass": m, "pos": p, "vel": v, "dump_steps": np.array([0, 1, 9])}
    for s in (0, 1, 9):
        d[f"pos_after_{s}"] = rd(out, f"pos_{s}.bin", (1000, 2))
        d[f"vel_after_{s}"] = rd(out, f"vel_{s}.bin", (1000, 2))
        d[f"forces_{s}"] = rd(out, f"forces_{s}.bin", (1000, 2))
    d["tree_0"] = tree_of(out, 0)
    np.savez_compressed(os.path.join(GOLD, "ref_ma2_1000.npz"), **d)

    # -- (7) main_approach_1.cpp (direct sum), n set to 1024, 100 steps ------------------------
    out = run_ref("ref_ma1_1024", init1024, 100, [0, 9, 99])
    d = {"dump_steps": np.array([0, 9, 99])}
    for s in (0, 9, 99):
        d[f"pos_after_{s}"] = rd(out, f"pos_{s}.bin", (1024, 2))
        d[f"vel_after_{s}"] = rd(out, f"vel_{s}.bin", (1024, 2))
        d[f"forces_{s}"] = rd(out, f"forces_{s}.bin", (1024, 2))
    np.savez_compressed(os.path.join(GOLD, "ref_ma1_1024.npz"), **d)

    for f in sorted(os.listdir(GOLD)):
        p = os.path.join(GOLD, f)
        if os.path.isfile(p):
            print(f"{f:32s} {os.path.getsize(p) / 1024:8.1f} KiB")


if __name__ == "__main__":
    main()
