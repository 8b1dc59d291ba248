"""pytest configuration: markers, repo root on sys.path, shared fixtures."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver at round end)")


@pytest.fixture(scope="session")
def gold():
    """name -> lazily loaded npz from tests/golden/."""
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLD, name + ".npz"))
        return cache[name]

    return get


@pytest.fixture(scope="session")
def init1024():
    """(mass, pos, vel): the first 1,024 lines of the reference's shipped init files."""
    d = os.path.join(GOLD, "init1024")
    return (np.loadtxt(os.path.join(d, "masses_init.txt")),
            np.loadtxt(os.path.join(d, "positions_init.txt")),
            np.loadtxt(os.path.join(d, "velocities_init.txt")))
