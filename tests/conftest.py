import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip():
    """The initialised HIP library; GPU tests fail loudly if it is missing."""
    import machisplin_amd
    machisplin_amd.init()
    return machisplin_amd


def synth_stations(n, seed, geom=None, distinct_cells=True):
    """SURVEY.md section 8d generator: stations on distinct cell centres of `geom`
    (or uniform in a 2x2 degree box), residual r = sin(6u) cos(5v) + 0.1 N(0,1)."""
    rng = np.random.default_rng(seed)
    if geom is not None:
        cells = rng.choice(geom.nrow * geom.ncol, size=n, replace=False)
        rows, cols = np.divmod(cells, geom.ncol)
        xy = np.column_stack([geom.x_from_col(cols), geom.y_from_row(rows)])
    else:
        xy = np.column_stack([rng.uniform(-78.0, -76.0, n), rng.uniform(-7.0, -5.0, n)])
    u = (xy - xy.min(0)) / (xy.max(0) - xy.min(0))
    y = np.sin(6 * u[:, 0]) * np.cos(5 * u[:, 1]) + 0.1 * rng.standard_normal(n)
    return xy, y
