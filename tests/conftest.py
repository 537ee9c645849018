import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
os.environ.setdefault("SIXDGS_RANDOM_BACKBONE", "1")     # no DINOv2 weights can be downloaded here: the tests opt in to a random ViT-S/14


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def syn():
    return importlib.import_module("6dgs_amd.synthetic")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"))
        return cache[name]

    return load


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def quadricell_tie_cells(eid, points):
    """Cells whose angle is an exact mathematical tie of the reference's arc-length look-up.

    The table of quadricell.py:255-281 is sampled with the ring's own cell step, so for rings whose
    cell count (or half of it) divides 999 -- 3, 6, 9, 18 cells -- table[333 m] equals 2*pi*m/3
    exactly, the very angle of cell j = m*cells/3.  `table < theta` is then decided by the last-ulp
    rounding of sin/cos in whatever library computed the table: the reference's own choice there is
    noise.  Returns a boolean mask of those cells (ring = run of equal (ellipsoid, z))."""
    eid = np.asarray(eid)
    z = np.asarray(points)[:, 2]
    n = eid.shape[0]
    start = np.ones(n, bool)
    start[1:] = (eid[1:] != eid[:-1]) | (z[1:] != z[:-1])
    ring_id = np.cumsum(start) - 1
    first = np.nonzero(start)[0]
    cells = np.diff(np.append(first, n))
    j = np.arange(n) - first[ring_id]
    c = cells[ring_id]
    return (j > 0) & ((3 * j) % c == 0) & np.isin(c, (3, 6, 9, 18))
