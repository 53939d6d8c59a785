"""Raster wire formats on both sides of the hot path.

``read_raster`` / ``read_stack`` <-> ``terra::rast(files)``: GeoTIFF covariates (INT2S / FLT4S /
FLT8S, LZW or deflate, strips or tiles, NoData tag, geo tags or ``.tfw`` sidecar) decoded on
host threads and streamed into device planes.  ``write_geotiff`` <-> ``terra::writeRaster`` as
``machisplin.write.geotiff`` calls it (V73:1011,1020): one FLT4S GeoTIFF per layer.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np

from . import _lib
from .raster import Geometry, RasterStack

_NP = {(_lib.I16): np.int16, (_lib.F32): np.float32, (_lib.F64): np.float64}
_HOST_NP = {(1, 8): np.uint8, (2, 8): np.int8, (1, 16): np.uint16, (2, 16): np.int16, (1, 32): np.uint32,
            (2, 32): np.int32, (3, 32): np.float32, (3, 64): np.float64, (1, 64): np.uint64, (2, 64): np.int64}


def tiff_info(path: str, ifd: int = 0) -> dict:
    info = _lib.TiffInfo()
    _lib.check(_lib.load().mhs_tiff_info_read(os.fsencode(path), ifd, C.byref(info)))
    return {f: getattr(info, f) for f, _ in info._fields_}


def world_file(path: str):
    """ESRI world file next to a raster (``alt.tfw`` for ``alt.tif``): xres, yres, xmin, ymax of the extent."""
    six = (C.c_double * 6)()
    _lib.check(_lib.load().mhs_tfw_read(os.fsencode(path), six))
    a, d, b, e, cx, cy = list(six)
    if d != 0.0 or b != 0.0:
        raise ValueError("rotated world files are not supported")
    return {"xres": a, "yres": -e, "xmin": cx - 0.5 * a, "ymax": cy - 0.5 * e}


def geometry_of(path: str, ifd: int = 0) -> Geometry | None:
    """Geometry from the GeoTIFF tags, else from a ``.tfw`` sidecar (``x.tif`` -> ``x.tfw``), else None.
    Overview levels (ifd > 0) scale the level-0 cell size by the size ratio."""
    i0 = tiff_info(path, 0)
    ii = tiff_info(path, ifd) if ifd else i0
    if i0["has_geo"]:
        xmin, ymax, xres, yres = i0["xmin"], i0["ymax"], i0["xres"], i0["yres"]
    else:
        base = path[:-4] if path.lower().endswith(".ovr") else path
        tfw = os.path.splitext(base)[0] + ".tfw"
        if not os.path.exists(tfw):
            return None
        w = world_file(tfw)
        xmin, ymax, xres, yres = w["xmin"], w["ymax"], w["xres"], w["yres"]
        if path.lower().endswith(".ovr"):  # .ovr level 0 is the first overview: half the base resolution
            xres, yres = xres * 2, yres * 2
    if ifd:
        xres, yres = xres * i0["width"] / ii["width"], yres * i0["height"] / ii["height"]
    return Geometry(xmin, ymax, xres, yres, ii["height"], ii["width"])


def read_host(path: str, ifd: int = 0) -> np.ndarray:
    """Decode one image directory into a numpy array of the file's native sample type (no GPU)."""
    i = tiff_info(path, ifd)
    out = np.empty((i["height"], i["width"]), dtype=_HOST_NP[(i["sample_format"], i["bits"])])
    _lib.check(_lib.load().mhs_tiff_read_host(os.fsencode(path), ifd, out.ctypes.data, out.nbytes))
    return out


def read_raster(path: str, ifd: int = 0):
    """terra::rast(path): (geometry or None, device plane (nrow, ncol) of int16 / float32 / float64, nodata)."""
    import torch
    i = tiff_info(path, ifd)
    if i["dtype"] < 0:
        raise ValueError(f"{path}: sample type ({i['sample_format']}, {i['bits']} bits) has no device plane type")
    dev = torch.device("cuda", _lib.init())
    tdt = {_lib.I16: torch.int16, _lib.F32: torch.float32, _lib.F64: torch.float64}[i["dtype"]]
    plane = torch.empty((i["height"], i["width"]), dtype=tdt, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(_lib.lib().mhs_tiff_read_dev(os.fsencode(path), ifd, plane.data_ptr(), plane.stride(0), st))
    return geometry_of(path, ifd), plane, i["nodata"]


def read_stack(paths, ifd: int = 0, geom: Geometry | None = None) -> RasterStack:
    """rast_stack's covariate layers from one GeoTIFF per layer (same grid, same sample type)."""
    import torch
    planes, nodata, g0 = [], float("nan"), geom
    for p in paths:
        g, pl, nd = read_raster(p, ifd)
        g0 = g0 or g
        if planes and (pl.shape != planes[0].shape or pl.dtype != planes[0].dtype):
            raise ValueError("all layers must share one grid and one sample type")
        if not math.isnan(nd):
            nodata = nd
        planes.append(pl)
    if g0 is None:
        raise ValueError("no georeference found (tags or .tfw); pass geom=")
    return RasterStack(g0, torch.stack(planes), nodata)


def write_geotiff(path: str, geom: Geometry, plane, nodata: float = float("nan"), compress: bool = True) -> None:
    """terra::writeRaster(layer, "<name>.tif") (V73:1011,1020): FLT4S strips, deflate; NaN -> nodata if given."""
    import torch
    g = geom.c_struct()
    comp = 8 if compress else 1
    if isinstance(plane, np.ndarray):
        data = np.ascontiguousarray(plane, dtype=np.float32)
        if not math.isnan(nodata):
            data = np.where(np.isnan(data), np.float32(nodata), data)
        _lib.check(_lib.load().mhs_tiff_write_f32_host(os.fsencode(path), C.byref(g), data.ctypes.data, float(nodata), comp))
        return
    if plane.dtype != torch.float64 or not plane.is_cuda or plane.dim() != 2 or plane.stride(1) != 1:
        raise ValueError("plane must be a 2-D float64 device tensor (or a numpy array)")
    st = torch.cuda.current_stream(plane.device).cuda_stream
    _lib.check(_lib.lib().mhs_tiff_write_f32_dev(os.fsencode(path), C.byref(g), plane.data_ptr(), plane.stride(0),
                                                 float(nodata), comp, st))
