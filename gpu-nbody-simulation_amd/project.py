"""The reference program's entry points under their own names, driving the HIP engine.

`runSimulationGpu` mirrors project.cu:918-1024 (same arguments plus the former compile-time
constants as keywords; same files written: quadtree_init_gpu.txt at step 0 and
quadtree_final_gpu.txt at the last step, each BEFORE that step's update, project.cu:962-965);
`main` mirrors project.cu:1049-1105 and prints exactly the two timing lines the scaling scripts
and plot_first_scale.py:55-59 parse:

    GPU total computation took <int> milliseconds.
    GPU parallel computation took <int> microseconds.

    python -m gpu_nbody_simulation_amd.project -DN_BODIES=1024 -DN_THREADS=1024 -DN_SIMULATIONS=100

(-D options are accepted in the reference's own spelling, so `nvcc -DN_BODIES=... project.cu`
lines of the scaling scripts translate one to one.)
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

from .engine import BarnesHutEngine, BhConfig, Precision
from .textio import loadSimulationDataFromText

# project.cu:27-35, 60-61
G = 6.67e-11
DELTA_T = 1.0
THETA = 5e-1
QUADTREE_MAX_DEPTH = 10
LOWER_M, HIGHER_M = 1e-1, 5e-1
LOWER_P, HIGHER_P = -1e-1, 1e-1
LOWER_V, HIGHER_V = -1e-4, 1e-4


def initializeCpu(n_bodies: int, seed: int = 0, save_to_file: bool = False):
    """initializeCpu (project.cu:298-302): log-uniform masses, uniform positions/velocities.
    numpy's PCG64 replaces rand(); the reference seeds from time() and is not reproducible."""
    rng = np.random.default_rng(seed)
    masses = 10.0 ** (np.log10(LOWER_M) + rng.random(n_bodies) * (np.log10(HIGHER_M) - np.log10(LOWER_M)))
    positions = LOWER_P + rng.random((n_bodies, 2)) * (HIGHER_P - LOWER_P)
    velocities = LOWER_V + rng.random((n_bodies, 2)) * (HIGHER_V - LOWER_V)
    if save_to_file:
        from .textio import save_init_files
        save_init_files(masses, positions, velocities)
        print("Masses saved to masses_init.txt")
        print("Vectors saved to positions_init.txt")
        print("Vectors saved to velocities_init.txt")
    return masses, positions, velocities


def initializeGpu(n_bodies: int, seed: int = 0, precision: Precision = Precision.F64_EXACT,
                  save_to_file: bool = False, device: int = 0):
    """initializeGpu (project.cu:304-341): bodies generated on the device with the reference's
    ranges (project.cu:30-35), reproducible for a given seed (the reference seeds cuRAND from
    time(NULL), project.cu:323).  save_to_file writes the three init files as initializeCpu's
    save_to_file branch does (project.cu:298-302)."""
    with BarnesHutEngine(BhConfig(capacity=max(n_bodies, 1), precision=precision, device=device)) as eng:
        eng.initialize(n_bodies, seed, "box", LOWER_M, HIGHER_M, LOWER_P, HIGHER_P, LOWER_V, HIGHER_V)
        positions, velocities = eng.download()
        masses = eng.masses()
    if save_to_file:
        from .textio import save_init_files
        save_init_files(masses, positions, velocities)
    return masses, positions, velocities


def runSimulationGpu(masses, positions, velocities, n_simulations: int, *, n_threads: int = 0,
                     theta: float = THETA, g: float = G, delta_t: float = DELTA_T,
                     max_depth: int = QUADTREE_MAX_DEPTH, precision: Precision = Precision.F64_EXACT,
                     reference_compat: bool = True, out_dir: str = ".", device: int = 0,
                     positions_file: str | None = None):
    """Returns (final_positions, final_velocities, gpu_parallel_duration_us).

    positions is NOT modified in place (the reference updates its by-reference argument,
    project.cu:918, 1010; the caller gets the same values as the first return value).
    positions_file: also write the trajectory as runSimulationCpu does into positions_cpu.txt
    (savePositions, project.cu:855-863, 876, 909: `t i x y ` per body, before the first step and
    after every step); this downloads the positions every step."""
    n = len(masses)
    # both files are opened (truncated) up front, as the reference's ofstreams are (project.cu:928-929)
    init_path = os.path.join(out_dir, "quadtree_init_gpu.txt")
    final_path = os.path.join(out_dir, "quadtree_final_gpu.txt")
    open(init_path, "w").close()
    open(final_path, "w").close()

    gpu_parallel_us = 0.0
    with BarnesHutEngine(BhConfig(capacity=max(n, 1), theta=theta, G=g, dt=delta_t, max_depth=max_depth,
                                  precision=precision, reference_compat=reference_compat,
                                  device=device, n_threads=n_threads)) as eng:
        eng.upload(positions, velocities, masses)
        traj = None
        absolute_t = 0.0
        if positions_file is not None:
            traj = open(os.path.join(out_dir, positions_file) if not os.path.isabs(positions_file) else positions_file, "w")
            _write_frame(traj, absolute_t, np.asarray(positions, dtype=np.float64))

        def advance(k):
            nonlocal gpu_parallel_us, absolute_t
            if k <= 0:
                return
            if traj is None:
                eng.step(k)
                gpu_parallel_us += eng.stats().last_step_ms * k * 1e3
                return
            for _ in range(k):
                eng.step(1)
                gpu_parallel_us += eng.stats().last_step_ms * 1e3
                absolute_t += delta_t
                _write_frame(traj, absolute_t, eng.download()[0])

        step = 0
        while step < n_simulations:
            if step == 0:
                eng.build_tree()
                eng.write_quadtree_file(init_path)
                advance(1)
                step += 1
            elif step == n_simulations - 1:
                eng.build_tree()
                eng.write_quadtree_file(final_path)
                advance(1)
                step += 1
            else:
                k = n_simulations - 1 - step
                advance(k)
                step += k
        pos, vel = eng.download()
        if traj is not None:
            traj.close()
    return pos, vel, gpu_parallel_us


def _write_frame(f, t: float, pos) -> None:
    """One savePositions call (project.cu:855-863): std::to_string formatting is "%f"."""
    f.write("".join("%f %d %f %f \n" % (t, i, x, y) for i, (x, y) in enumerate(pos)))


def _parse(argv):
    ap = argparse.ArgumentParser(prog="project", description=__doc__,
                                 formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-D", action="append", default=[], metavar="NAME=VALUE",
                    help="N_BODIES / N_THREADS / N_SIMULATIONS, as on the reference's nvcc line")
    ap.add_argument("--n-bodies", type=int)
    ap.add_argument("--n-threads", type=int)
    ap.add_argument("--n-simulations", type=int)
    ap.add_argument("--init", choices=["auto", "files", "random", "gpu"], default="auto",
                    help="files: loadSimulationDataFromText from the CWD; random: initializeCpu; "
                         "gpu: initializeGpu (on-device generator); auto: files when masses_init.txt "
                         "exists, else gpu -- the reference as shipped calls initializeGpu")
    ap.add_argument("--positions-file", default=None, help="also write the trajectory (savePositions format)")
    ap.add_argument("--save-init", action="store_true", help="write the three init files after initialisation")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--precision", choices=["f64", "f32"], default="f64")
    ap.add_argument("--max-depth", type=int, default=QUADTREE_MAX_DEPTH)
    ap.add_argument("--theta", type=float, default=THETA)
    ap.add_argument("--no-compat", action="store_true", help="bucket leaves instead of the depth-cap artefact")
    ap.add_argument("-o", dest="ignored_output", help="accepted and ignored (nvcc line compatibility)")
    ap.add_argument("source", nargs="?", help="accepted and ignored (nvcc line compatibility)")
    a = ap.parse_args(argv)
    # project.cu:1-11.  N_THREADS: the reference's default is 1,024 CUDA threads striding over the bodies; here it
    # caps the bodies walked at a time ONLY when given (-DN_THREADS=... / --n-threads, as the scaling scripts do):
    # unset, a step walks all bodies in one launch.
    macros = {"N_BODIES": 1000 * 40, "N_THREADS": 0, "N_SIMULATIONS": 10}
    for d in a.D:
        k, _, v = d.partition("=")
        if k not in macros:
            ap.error(f"unknown macro {k}")
        try:
            macros[k] = _macro_value(v)
        except ValueError:
            ap.error(f"-D{k}={v}: expected an integer or a product of integers (the reference writes `1000 * 40`)")
    if a.n_bodies is not None:
        macros["N_BODIES"] = a.n_bodies
    if a.n_threads is not None:
        macros["N_THREADS"] = a.n_threads
    if a.n_simulations is not None:
        macros["N_SIMULATIONS"] = a.n_simulations
    return a, macros


def _macro_value(text: str) -> int:
    """Value of a -D macro as the reference writes them (project.cu:1-11): an integer literal or a
    product of integer literals such as `1000 * 40`, optionally parenthesised.  Nothing is evaluated."""
    t = text.strip()
    while t.startswith("(") and t.endswith(")"):
        t = t[1:-1].strip()
    value = 1
    for factor in t.split("*"):
        value *= int(factor.strip(), 10)               # ValueError on anything but a decimal integer
    return value


def main(argv=None) -> int:
    a, mac = _parse(sys.argv[1:] if argv is None else argv)
    n = mac["N_BODIES"]
    use_files = a.init == "files" or (a.init == "auto" and os.path.exists("masses_init.txt"))
    if use_files:
        masses, positions, velocities = loadSimulationDataFromText(
            "masses_init.txt", "positions_init.txt", "velocities_init.txt", n, N_BODIES=n)
    elif a.init == "random":
        masses, positions, velocities = initializeCpu(n, seed=a.seed, save_to_file=a.save_init)
    else:
        masses, positions, velocities = initializeGpu(
            n, seed=a.seed, precision=Precision.F64_EXACT if a.precision == "f64" else Precision.F32,
            save_to_file=a.save_init)

    start = time.perf_counter()
    _, _, gpu_parallel_us = runSimulationGpu(
        masses, positions, velocities, mac["N_SIMULATIONS"], n_threads=mac["N_THREADS"],
        theta=a.theta, max_depth=a.max_depth,
        precision=Precision.F64_EXACT if a.precision == "f64" else Precision.F32,
        reference_compat=not a.no_compat, positions_file=a.positions_file)
    duration_ms = int((time.perf_counter() - start) * 1e3)

    # project.cu:1090-1102, blank lines included
    sys.stdout.write("\n\n")
    sys.stdout.write("\n\n")
    sys.stdout.write(f"GPU total computation took {duration_ms} milliseconds.\n")
    sys.stdout.write("\n\n")
    sys.stdout.write(f"GPU parallel computation took {int(gpu_parallel_us)} microseconds.\n")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
