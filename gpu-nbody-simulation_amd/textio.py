"""The reference's text formats: the three init files in, quadtree_*.txt / positions_*.txt out.

loadSimulationDataFromText mirrors project.cu:103-161 (same argument order and meaning, same
failure messages); save_init_files mirrors the save_to_file branches of initializeMasses /
initializeVectors (project.cu:236-246, 269-281); parse_quadtree_file returns what
plot_quadtree.py:11-45 returns, so that our writer can be checked against the reference's consumer
contract without importing it.
"""
from __future__ import annotations

import re

import numpy as np

_FLOAT_PREFIX = re.compile(r"\s*[-+]?(?:(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?|inf(?:inity)?|nan)", re.I)


def _stod(line: str, filename: str) -> float:
    """std::stod: leading whitespace, longest numeric prefix, trailing text ignored."""
    m = _FLOAT_PREFIX.match(line)
    if not m:
        raise ValueError(f"stod: no conversion in file: {filename}")     # std::invalid_argument
    return float(m.group(0))


def loadSimulationDataFromText(massesFile: str, positionsFile: str, velocitiesFile: str,
                               n_bodies: int, N_BODIES: int | None = None, verbose: bool = True):
    """Returns (masses[n], positions[n,2], velocities[n,2]) as float64 -- project.cu:103-161.

    N_BODIES is the capacity the reference fixes at compile time (project.cu:1-3); when given,
    n_bodies > N_BODIES raises like the reference's std::out_of_range (project.cu:110-112).
    """
    if N_BODIES is not None and n_bodies > N_BODIES:
        raise IndexError("Requested number of bodies exceeds N_BODIES.")

    def open_or_fail(name):
        try:
            return open(name, "r")
        except OSError:
            raise RuntimeError("Failed to open file: " + name) from None

    masses = np.empty(n_bodies)
    with open_or_fail(massesFile) as f:
        for i in range(n_bodies):
            line = f.readline()
            if line == "":
                raise RuntimeError("Not enough mass entries in file: " + massesFile)
            masses[i] = _stod(line, massesFile)

    def load_vectors(name):
        out = np.empty((n_bodies, 2))
        with open_or_fail(name) as f:
            for i in range(n_bodies):
                line = f.readline()
                if line == "":
                    raise RuntimeError("Not enough vector entries in file: " + name)
                tok = line.split()
                for d in range(2):
                    try:
                        out[i, d] = float(tok[d])
                    except (IndexError, ValueError):
                        raise RuntimeError("Failed to parse vector component in file: " + name) from None
        return out

    positions = load_vectors(positionsFile)
    velocities = load_vectors(velocitiesFile)
    if verbose:
        print(f"Loaded {n_bodies} bodies from text files.")                # project.cu:160
    return masses, positions, velocities


def save_init_files(masses, positions, velocities, masses_file="masses_init.txt",
                    positions_file="positions_init.txt", velocities_file="velocities_init.txt",
                    exact: bool = False) -> None:
    """Write the three init files.  Default formatting is the reference's (`ofs << double`, i.e.
    "%g", 6 significant digits, project.cu:241-243, 274-278); exact=True writes round-trip repr
    instead, for inputs that must be read back bit for bit."""
    fmt = (lambda x: repr(float(x))) if exact else (lambda x: "%g" % x)
    with open(masses_file, "w") as f:
        for m in np.asarray(masses).reshape(-1):
            f.write(fmt(m) + "\n")
    for name, arr in ((positions_file, positions), (velocities_file, velocities)):
        with open(name, "w") as f:
            for x, y in np.asarray(arr).reshape(-1, 2):
                f.write(fmt(x) + " " + fmt(y) + "\n")


def save_positions(path: str, frames) -> None:
    """positions_*.txt as savePositions writes it (project.cu:855-863): `t i x y ` per line with
    std::to_string formatting ("%f").  frames: iterable of (time, positions[n,2])."""
    with open(path, "w") as f:
        for t, pos in frames:
            for i, (x, y) in enumerate(np.asarray(pos).reshape(-1, 2)):
                f.write("%f %d %f %f \n" % (t, i, x, y))


def parse_positions_file(path: str):
    """(times[T], positions[T, n, 2]) from a positions_*.txt trajectory (`t i x y ` per line, what
    save_positions / savePositions write and plot_2d.py:7-16 reads).  Frames must be complete."""
    rows = np.loadtxt(path, ndmin=2)
    if rows.size == 0:
        return np.zeros(0), np.zeros((0, 0, 2))
    body = rows[:, 1].astype(np.int64)
    n = int(body.max()) + 1
    if len(rows) % n or not np.array_equal(body, np.tile(np.arange(n), len(rows) // n)):
        raise ValueError(f"{path}: not a sequence of complete frames of {n} bodies")
    return rows[::n, 0].copy(), rows[:, 2:4].reshape(-1, n, 2).copy()


def plot_trajectories(path: str, png_path: str, max_bodies: int = 64) -> int:
    """Counterpart of the reference's plot_2d.py: one polyline per body (the first max_bodies of them),
    written to png_path.  Returns the number of bodies drawn."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    _, pos = parse_positions_file(path)
    nb = min(pos.shape[1], max_bodies)
    fig, ax = plt.subplots(figsize=(8, 8))
    for b in range(nb):
        ax.plot(pos[:, b, 0], pos[:, b, 1], marker="o", markersize=2, linewidth=0.8)
    ax.set_title("N-Body Problem Visualization")
    ax.set_xlabel("X Coordinate")
    ax.set_ylabel("Y Coordinate")
    ax.axhline(0, color="gray", linestyle="--", linewidth=0.5)
    ax.axvline(0, color="gray", linestyle="--", linewidth=0.5)
    ax.grid(True)
    fig.savefig(png_path, dpi=100)
    plt.close(fig)
    return nb


_OCC = re.compile(r"occupantIndex=(-?\d+)\s+occupantPos=\(([-0-9.e+]+),([-0-9.e+]+)\)")


def parse_quadtree_file(filename: str):
    """[(depth, x_min, x_max, y_min, y_max, total_mass, [(idx, x, y), ...]), ...] -- the tuple
    layout of plot_quadtree.py:11-45 (>= 6 whitespace tokens, then the occupant regex)."""
    out = []
    with open(filename) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            tok = line.split()
            if len(tok) < 6:
                continue
            occ = [(int(a), float(b), float(c)) for a, b, c in _OCC.findall(line)]
            out.append((int(tok[0]), float(tok[1]), float(tok[2]), float(tok[3]), float(tok[4]),
                        float(tok[5]), occ))
    return out
