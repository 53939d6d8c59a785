"""Host-side mirror of the reference's tile bookkeeping on the hot path: the Step-3 TPS
tile grid (V73:656-681), terra::crop windows, the mean mosaic + seam feathering + overlay
(V73:739-747, 760-895) and machisplin.tiles.create / machisplin.tiles.merge (V73:1165-1256,
1392-1548).  Integer windows come from the library's host functions (bit-exact, no GPU);
the per-cell blends run in HIP kernels.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .raster import Geometry


def crop_window(geom: Geometry, ext):
    """terra::crop(x, ext): (r0, r1, c0, c1) half-open window of `geom`; ext = (xmin, xmax, ymin, ymax)."""
    e = np.ascontiguousarray(np.asarray(ext, dtype=np.float64))
    w = np.empty(4, dtype=np.int64)
    g = geom.c_struct()
    _lib.check(_lib.load().mhs_crop_window(C.byref(g), e.ctypes.data, w.ctypes.data))
    return tuple(int(v) for v in w)


def step3_tile_windows(geom: Geometry, tile_edge: int = 1500, fit_overlap: float = 0.2, keep_overlap: float = 0.025):
    """V73:656-681 + the crops at V73:699,728.  Returns nRx, nCx, fit windows, keep windows
    (arrays n x 4: r0, r1, c0, c1), tiles numbered row-major from the south-west."""
    lib = _lib.load()
    g = geom.c_struct()
    nR, nC = C.c_int64(), C.c_int64()
    _lib.check(lib.mhs_step3_tile_windows(C.byref(g), tile_edge, fit_overlap, keep_overlap, C.byref(nR), C.byref(nC), None, None, 0))
    n = nR.value * nC.value
    fit = np.empty((n, 4), dtype=np.int64)
    keep = np.empty((n, 4), dtype=np.int64)
    _lib.check(lib.mhs_step3_tile_windows(C.byref(g), tile_edge, fit_overlap, keep_overlap, C.byref(nR), C.byref(nC),
                                          fit.ctypes.data, keep.ctypes.data, n))
    return nR.value, nC.value, fit, keep


def cells_from_xy(geom: Geometry, xy):
    """terra::cellFromXY: rows, cols of the cells holding the points (-1 outside)."""
    xy = np.asfortranarray(np.asarray(xy, dtype=np.float64).reshape(-1, 2))
    rows = np.empty(xy.shape[0], dtype=np.int64)
    cols = np.empty(xy.shape[0], dtype=np.int64)
    g = geom.c_struct()
    _lib.check(_lib.load().mhs_cells_from_xy(C.byref(g), xy.ctypes.data, xy.shape[0], rows.ctypes.data, cols.ctypes.data))
    return rows, cols


def extract(plane, rows, cols) -> np.ndarray:
    """terra::extract(r, xy) given the cells: gather from a 2-D float64 device plane."""
    import torch
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    cols = np.ascontiguousarray(cols, dtype=np.int64)
    out = np.empty(rows.size)
    st = torch.cuda.current_stream(plane.device).cuda_stream
    _lib.check(_lib.lib().mhs_gather_cells_dev(plane.data_ptr(), plane.stride(0), rows.ctypes.data, cols.ctypes.data,
                                               rows.size, out.ctypes.data, st))
    return out


def mosaic_feather(geom: Geometry, nRx: int, nCx: int, windows, tiles, merge_mode: bool = False, out=None,
                   return_seams: bool = False):
    """Mean mosaic of the tiles, linear cross-fade of every seam strip, strips laid over the
    mosaic (first non-NA): Step 3 mosaic + Step 4 of machisplin.mltps, or tiles.merge with
    merge_mode.  `tiles[h]` is a contiguous float64 device tensor covering windows[h]."""
    import torch
    windows = np.ascontiguousarray(windows, dtype=np.int64).reshape(-1, 4)
    n = nRx * nCx
    if len(tiles) != n or windows.shape[0] != n:
        raise ValueError("need nRx*nCx tiles and windows")
    dev = tiles[0].device
    for h, t in enumerate(tiles):
        w = windows[h]
        if t.dtype != torch.float64 or not t.is_cuda or not t.is_contiguous() or tuple(t.shape) != (w[1] - w[0], w[3] - w[2]):
            raise ValueError(f"tile {h} must be a contiguous float64 device tensor of its window's shape")
    if out is None:
        out = torch.empty((geom.nrow, geom.ncol), dtype=torch.float64, device=dev)
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in tiles])
    ns = C.c_int64()
    _lib.check(_lib.load().mhs_seam_count(nRx, nCx, C.byref(ns)))
    seams = np.full((max(ns.value, 1), 4), -1, dtype=np.int64)
    g = geom.c_struct()
    st = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(_lib.lib().mhs_mosaic_feather_dev(C.byref(g), nRx, nCx, windows.ctypes.data, ptrs, int(merge_mode),
                                                 out.data_ptr(), out.stride(0), seams.ctypes.data, st))
    return (out, seams[:ns.value]) if return_seams else out


def tiles_create(geom: Geometry, int_values_xy, out_ncol: int = 3, out_nrow: int = 3, feather_d: float = 50):
    """machisplin.tiles.create (V73:1165-1256): per tile its extent box, the crop window of the
    rasters and the indices of the stations inside the box (borders inclusive).  Tiles are
    ordered row-major from the south-west; each is an independent mltps run (one GPU each)."""
    n = out_ncol * out_nrow
    boxes = np.empty((n, 4))
    win = np.empty((n, 4), dtype=np.int64)
    g = geom.c_struct()
    _lib.check(_lib.load().mhs_tiles_create_windows(C.byref(g), out_ncol, out_nrow, float(feather_d),
                                                    boxes.ctypes.data, win.ctypes.data))
    xy = np.asarray(int_values_xy, dtype=np.float64)
    dat = [np.flatnonzero((xy[:, 0] >= b[0]) & (xy[:, 0] <= b[1]) & (xy[:, 1] >= b[2]) & (xy[:, 1] <= b[3])) for b in boxes]
    return {"e.ext": boxes, "win": win, "dat": dat, "nC": out_ncol, "nR": out_nrow,
            "geom": [geom.window(*[int(v) for v in w]) for w in win]}


def tiles_merge(geom: Geometry, windows, rast_in, in_ncol: int = 2, in_nrow: int = 3, out=None):
    """machisplin.tiles.merge (V73:1392-1548): feather-merge the per-tile finals."""
    return mosaic_feather(geom, in_nrow, in_ncol, windows, rast_in, merge_mode=True, out=out)
