"""Raster geometry and device-resident raster planes (the slice of terra's SpatRaster
the hot path touches: ext/res/dim, cell-centre coordinates, crop windows).

Cell order is terra's: row-major from the north-west cell (V73:128-133).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import _lib


@dataclass(frozen=True)
class Geometry:
    """terra::ext + dim of a SpatRaster (what ``terra::rast(rb)`` keeps, V73:726)."""
    xmin: float
    ymax: float
    xres: float
    yres: float
    nrow: int
    ncol: int

    @property
    def xmax(self) -> float:
        return self.xmin + self.ncol * self.xres

    @property
    def ymin(self) -> float:
        return self.ymax - self.nrow * self.yres

    @property
    def ncell(self) -> int:
        return self.nrow * self.ncol

    def ext(self):
        """terra::ext order: (xmin, xmax, ymin, ymax)."""
        return (self.xmin, self.xmax, self.ymin, self.ymax)

    def x_from_col(self, col):
        """terra::xFromCol (0-based col): cell-centre longitude."""
        return self.xmin + (np.asarray(col, dtype=np.float64) + 0.5) * self.xres

    def y_from_row(self, row):
        """terra::yFromRow (0-based row, from the north edge): cell-centre latitude."""
        return self.ymax - (np.asarray(row, dtype=np.float64) + 0.5) * self.yres

    def col_from_x(self, x):
        """terra::colFromX (0-based; -1 outside).  The east edge belongs to the last column."""
        x = np.asarray(x, dtype=np.float64)
        col = np.floor((x - self.xmin) / self.xres).astype(np.int64)
        col = np.where(x == self.xmax, self.ncol - 1, col)
        return np.where((x < self.xmin) | (x > self.xmax), -1, col)

    def row_from_y(self, y):
        """terra::rowFromY (0-based; -1 outside).  The south edge belongs to the last row."""
        y = np.asarray(y, dtype=np.float64)
        row = np.floor((self.ymax - y) / self.yres).astype(np.int64)
        row = np.where(y == self.ymin, self.nrow - 1, row)
        return np.where((y < self.ymin) | (y > self.ymax), -1, row)

    def window(self, r0: int, r1: int, c0: int, c1: int) -> "Geometry":
        """Geometry of the sub-raster rows [r0,r1) x cols [c0,c1)."""
        return Geometry(self.xmin + c0 * self.xres, self.ymax - r0 * self.yres, self.xres,
                        self.yres, r1 - r0, c1 - c0)

    def c_struct(self) -> "_lib.Grid":
        return _lib.Grid(self.xmin, self.ymax, self.xres, self.yres, self.nrow, self.ncol)


class RasterStack:
    """``rast_stack``'s covariate layers (V73:138) resident in HBM: a (C, nrow, ncol) device
    tensor of float64 / float32 / int16 plus the NoData value (INT2S rasters use -32768).
    LONG and LAT are not stored: the kernels generate them from the geometry."""

    _DT = None

    def __init__(self, geom: Geometry, planes, nodata: float = float("nan")):
        import torch
        if RasterStack._DT is None:
            RasterStack._DT = {torch.float64: _lib.F64, torch.float32: _lib.F32, torch.int16: _lib.I16}
        if isinstance(planes, np.ndarray):
            planes = torch.from_numpy(np.ascontiguousarray(planes))
        if planes.dim() != 3 or tuple(planes.shape[1:]) != (geom.nrow, geom.ncol):
            raise ValueError("planes must be (C, nrow, ncol) matching the geometry")
        if planes.dtype not in RasterStack._DT:
            raise ValueError("planes must be float64, float32 or int16")
        dev = torch.device("cuda", _lib.init())
        self.planes = planes.to(dev).contiguous()
        self.geom = geom
        self.nodata = float(nodata)

    @property
    def n_layers(self) -> int:
        return self.planes.shape[0]

    def c_struct(self) -> "_lib.Stack":
        p = self.planes
        return _lib.Stack(p.data_ptr(), p.shape[0], RasterStack._DT[p.dtype], p.stride(0), p.stride(1),
                          self.nodata)
