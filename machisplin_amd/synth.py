"""Deterministic synthetic inputs of the BASELINE.json shapes:
grid geometry of the bundled rasters (1/1200 degree cells, NW origin -78, -5), stations on
distinct cell centres, and the TPS-only residual  r = sin(6u) cos(5v) + 0.1 N(0,1).
Used by bench.py, __graft_entry__.smoke() and the tests; no reference data involved."""
from __future__ import annotations

import numpy as np

from .raster import Geometry

BASE_SEED = 20251017


def grid(nrow: int, ncol: int) -> Geometry:
    return Geometry(-78.0, -5.0, 1.0 / 1200.0, 1.0 / 1200.0, nrow, ncol)


def stations(geom: Geometry, n: int, seed: int):
    """n stations on distinct cell centres (knots are cell centres, V73:128-133,145);
    returns xy (n x 2: LONG, LAT), the cell rows/cols, and unit-square coordinates."""
    rng = np.random.default_rng(seed)
    cells = rng.choice(geom.ncell, size=n, replace=False)
    rows, cols = np.divmod(cells, geom.ncol)
    xy = np.column_stack([geom.x_from_col(cols), geom.y_from_row(rows)])
    uv = np.column_stack([(cols + 0.5) / geom.ncol, (rows + 0.5) / geom.nrow])
    return xy, rows, cols, uv


def tps_residual(uv: np.ndarray, seed: int) -> np.ndarray:
    rng = np.random.default_rng(seed + 1)
    return np.sin(6 * uv[:, 0]) * np.cos(5 * uv[:, 1]) + 0.1 * rng.standard_normal(uv.shape[0])
