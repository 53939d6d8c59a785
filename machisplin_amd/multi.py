"""Several MI355X from ONE host process, through the library's own multi-device entry points
(include/machisplin_hip.h, section "several devices"; csrc/multi.hip) -- the path the single-threaded R host takes.
:mod:`machisplin_amd.sharded` is the other route to the same arithmetic: one PROCESS per GPU under
``torch.distributed`` (what ``bench.py --gpus N`` runs when the driver launches it with torchrun).

Two drivers, as in the reference's own decomposition:
  * :func:`mltps_grid_multi` / :class:`MultiStack` -- machisplin.mltps Steps 2-5 (V73:442-930) for one response layer with
    the grid cut into row bands over the device slots;
  * :func:`tiles_units_multi` -- machisplin.tiles.create -> machisplin.mltps per (tile, layer) -> machisplin.tiles.merge
    (README.md:157-215 of the reference, V73:1165-1256, 1392-1548).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .raster import Geometry

_DT = {np.dtype(np.float64): _lib.F64, np.dtype(np.float32): _lib.F32, np.dtype(np.int16): _lib.I16}
COLLECTIVE = {0: "none", 1: "rccl-all-gather", 2: "peer-copies"}


def init_devices(n: int, device_ids=None):
    """mhs_init_devices: device slots 0..n-1 on the physical devices `device_ids` (default 0..n-1; ids may repeat --
    several slots on one GPU exercise every multi-device code path on a one-GPU box)."""
    lib = _lib.load()
    ids = None if device_ids is None else (C.c_int * n)(*[int(d) for d in device_ids])
    _lib.check(lib.mhs_init_devices(int(n), ids))
    dev0 = 0 if device_ids is None else int(device_ids[0])
    if _lib._inited_device is None:
        import atexit
        atexit.register(_lib._shutdown)
    _lib._inited_device = dev0
    return device_slots()


def device_slots():
    n = C.c_int()
    ids = (C.c_int * 16)()
    _lib.check(_lib.load().mhs_device_slots(C.byref(n), ids))
    return [int(ids[k]) for k in range(n.value)]


def trim():
    """mhs_multi_trim: release the device buffers, arenas and pinned rings the host-plane calls keep between calls."""
    _lib.check(_lib.lib().mhs_multi_trim())


def plan_row_bands(nrow: int, n_slots: int, slot0_share: float | None = None):
    """The row bands mhs_multi_stack_create cuts (host only, no GPU): ([(r0, r1)] per slot, band, lead)."""
    r0 = np.zeros(n_slots, dtype=np.int64)
    r1 = np.zeros(n_slots, dtype=np.int64)
    band, lead = C.c_int64(), C.c_int64()
    _lib.check(_lib.load().mhs_plan_row_bands(int(nrow), int(n_slots), float("nan") if slot0_share is None else float(slot0_share),
                                              r0.ctypes.data, r1.ctypes.data, C.byref(band), C.byref(lead)))
    return [(int(a), int(b)) for a, b in zip(r0, r1)], band.value, lead.value


def _host_stack(planes, nodata):
    planes = np.ascontiguousarray(planes)
    if planes.ndim != 3 or planes.dtype not in _DT:
        raise ValueError("planes must be a (C, nrow, ncol) float64 / float32 / int16 array in host memory")
    st = _lib.Stack(planes.ctypes.data, planes.shape[0], _DT[planes.dtype], planes.shape[1] * planes.shape[2], planes.shape[2],
                    float(nodata))
    return planes, st


def _members(models, weights):
    hs = (C.c_void_p * len(models))(*[m._h for m in models])
    ws = (C.c_double * len(models))(*[float(w) for w in weights])
    return hs, ws


def _info_dict(info: _lib.MltpsInfo):
    n = info.n_slots
    return {"rsq_model": info.rsq_model, "rsq_final": info.rsq_final, "lambda": info.lambda_, "n_knots": info.n_knots,
            "tiles": (info.tiles_rows, info.tiles_cols), "used_tps": bool(info.used_tps), "n_slots": n,
            "collective": COLLECTIVE.get(info.collective, "?"),
            "bands": [(info.band_r0[k], info.band_r1[k]) for k in range(n)],
            "band_ms": [info.band_ms[k] for k in range(n)], "tiles_ms": [info.tiles_ms[k] for k in range(n)],
            "fit_ms": info.fit_ms, "step_ms": info.step_ms, "upload_ms": info.upload_ms, "download_ms": info.download_ms,
            "suggested_slot0_share": info.suggested_slot0_share,
            "tiles_pulled_bytes": [info.tiles_pulled_bytes[k] for k in range(n)], "tiles_owned": [info.tiles_owned[k] for k in range(n)]}


class MultiStack:
    """rast_stack's covariate layers cut into row bands, band k resident on device slot k (mhs_multi_stack)."""

    def __init__(self, geom: Geometry, planes, nodata: float = float("nan"), slot0_share: float | None = None):
        planes, st = _host_stack(planes, nodata)
        if tuple(planes.shape[1:]) != (geom.nrow, geom.ncol):
            raise ValueError("planes must match the geometry")
        self.geom = geom
        g = geom.c_struct()
        h = C.c_void_p()
        _lib.check(_lib.lib().mhs_multi_stack_create(C.byref(g), C.byref(st), float("nan") if slot0_share is None else float(slot0_share),
                                                     C.byref(h)))
        self._h = h

    def bands(self):
        n = C.c_int()
        r0 = (C.c_int64 * 16)()
        r1 = (C.c_int64 * 16)()
        _lib.check(_lib.load().mhs_multi_stack_bands(self._h, C.byref(n), r0, r1))
        return [(int(r0[k]), int(r1[k])) for k in range(n.value)]

    def step(self, models, weights, wt_total, X, resp, tile_edge: int | None = None, lambda_=None, gcv_mode: str = "fields",
             gather: bool = False):
        """machisplin.mltps Steps 2-5 on the resident bands (mhs_mltps_grid_multi_dev); returns the info dict."""
        X = np.asfortranarray(np.asarray(X, dtype=np.float64))
        resp = np.ascontiguousarray(resp, dtype=np.float64)
        hs, ws = _members(models, weights)
        info = _lib.MltpsInfo()
        mode = {"fields": _lib.GCV_FIELDS, "converged": _lib.GCV_CONVERGED}[gcv_mode]
        _lib.check(_lib.lib().mhs_mltps_grid_multi_dev(hs, ws, len(models), float(wt_total), self._h, X.ctypes.data, resp.ctypes.data,
                                                       X.shape[0], 0 if tile_edge is None else int(tile_edge),
                                                       float("nan") if lambda_ is None else float(lambda_), mode, int(bool(gather)),
                                                       C.byref(info)))
        return _info_dict(info)

    def download(self) -> np.ndarray:
        out = np.empty((self.geom.nrow, self.geom.ncol))
        _lib.check(_lib.lib().mhs_multi_final_download(self._h, out.ctypes.data))
        return out

    def final_dev(self, slot: int):
        """(band pointer, r0, r1, gathered-grid pointer or None) of the last step on `slot` (raw device addresses)."""
        band, full = C.c_void_p(), C.c_void_p()
        r0, r1 = C.c_int64(), C.c_int64()
        _lib.check(_lib.load().mhs_multi_final_dev(self._h, int(slot), C.byref(band), C.byref(r0), C.byref(r1), C.byref(full)))
        return band.value, r0.value, r1.value, full.value

    def gathered(self, slot: int) -> np.ndarray:
        """The stitched grid as slot `slot` holds it after a step with gather=True (copied to the host)."""
        _, _, _, full = self.final_dev(slot)
        if not full:
            raise RuntimeError("the last step did not gather")
        out = np.empty((self.geom.nrow, self.geom.ncol))
        hip = C.CDLL(None)
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        rc = hip.hipMemcpy(out.ctypes.data, full, out.nbytes, 2)      # hipMemcpyDeviceToHost
        if rc != 0:
            raise RuntimeError(f"hipMemcpy failed: {rc}")
        return out

    def free(self):
        # also __del__: at interpreter shutdown the module global may already be None, and after the library's atexit
        # shutdown the handle's device memory is gone with its device -- nothing to free then
        if getattr(self, "_h", None) is None:
            return
        if _lib is not None and getattr(_lib, "_lib", None) is not None and getattr(_lib, "_inited_device", None) is not None:
            _lib._lib.mhs_multi_stack_free(self._h)
        self._h = None

    __del__ = free


def mltps_grid_multi(geom: Geometry, planes, nodata, models, weights, wt_total, X, resp, tile_edge: int | None = None, lambda_=None,
                     gcv_mode: str = "fields", slot0_share: float | None = None, out: np.ndarray | None = None):
    """Host planes in, host plane out, one library call (mhs_mltps_grid_multi -- what the R shim binds).  Returns
    (final plane, info).  `out`: a C-contiguous float64 (nrow, ncol) array to write the plane into (default: a new one)."""
    planes, st = _host_stack(planes, nodata)
    X = np.asfortranarray(np.asarray(X, dtype=np.float64))
    resp = np.ascontiguousarray(resp, dtype=np.float64)
    hs, ws = _members(models, weights)
    g = geom.c_struct()
    if out is None:
        out = np.empty((geom.nrow, geom.ncol))
    elif out.shape != (geom.nrow, geom.ncol) or out.dtype != np.float64 or not out.flags.c_contiguous:
        raise ValueError("out must be a C-contiguous float64 (nrow, ncol) array")
    info = _lib.MltpsInfo()
    mode = {"fields": _lib.GCV_FIELDS, "converged": _lib.GCV_CONVERGED}[gcv_mode]
    _lib.check(_lib.lib().mhs_mltps_grid_multi(hs, ws, len(models), float(wt_total), C.byref(g), C.byref(st), X.ctypes.data,
                                               resp.ctypes.data, X.shape[0], 0 if tile_edge is None else int(tile_edge),
                                               float("nan") if lambda_ is None else float(lambda_), mode,
                                               float("nan") if slot0_share is None else float(slot0_share), out.ctypes.data,
                                               C.byref(info)))
    return out, _info_dict(info)


def tiles_units_multi(geom: Geometry, planes, nodata, out_ncol: int, out_nrow: int, feather_d: float, units, n_layers: int,
                      tps: bool = True, tile_edge: int | None = 1500, lambda_=None, gcv_mode: str = "fields", merge_layers=None,
                      out=None):
    """machisplin.tiles.create -> mltps per (tile, layer) -> machisplin.tiles.merge over the device slots
    (mhs_tiles_units_multi).  `units[l][t]` = dict(models, weights, wt_total, X, resp) of tile t (row-major from the
    south-west) and layer l.  Returns (list of merged planes per layer -- None where `merge_layers` skips one --, rsq array
    (n_layers, n_tiles, 2), info dict).  `out`: a list of n_layers C-contiguous float64 (nrow, ncol) arrays (None where
    skipped) to write the merged planes into instead of new ones."""
    planes, st = _host_stack(planes, nodata)
    n_tiles = out_ncol * out_nrow
    keep = []                                   # numpy arrays and ctypes arrays the call reads
    arr = (_lib.Unit * (n_tiles * n_layers))()
    for l in range(n_layers):
        for t in range(n_tiles):
            u = units[l][t]
            hs, ws = _members(u["models"], u["weights"])
            X = np.asfortranarray(np.asarray(u["X"], dtype=np.float64))
            y = np.ascontiguousarray(u["resp"], dtype=np.float64)
            keep += [hs, ws, X, y]
            a = arr[l * n_tiles + t]
            a.models, a.weights, a.n_models = C.cast(hs, C.POINTER(C.c_void_p)), C.cast(ws, C.POINTER(C.c_double)), len(u["models"])
            a.wt_total, a.X, a.resp, a.n = float(u["wt_total"]), X.ctypes.data, y.ctypes.data, X.shape[0]
    want = list(range(n_layers)) if merge_layers is None else list(merge_layers)
    if out is None:
        outs = [np.empty((geom.nrow, geom.ncol)) if l in want else None for l in range(n_layers)]
    else:
        outs = [out[l] if l in want else None for l in range(n_layers)]
        for o in outs:
            if o is not None and (o.shape != (geom.nrow, geom.ncol) or o.dtype != np.float64 or not o.flags.c_contiguous):
                raise ValueError("out planes must be C-contiguous float64 (nrow, ncol) arrays")
    ptrs = (C.c_void_p * n_layers)(*[None if o is None else o.ctypes.data for o in outs])
    rsq = np.full((n_layers, n_tiles, 2), np.nan)
    info = _lib.UnitsInfo()
    g = geom.c_struct()
    mode = {"fields": _lib.GCV_FIELDS, "converged": _lib.GCV_CONVERGED}[gcv_mode]
    _lib.check(_lib.lib().mhs_tiles_units_multi(C.byref(g), C.byref(st), out_ncol, out_nrow, float(feather_d), n_layers, arr, int(bool(tps)),
                                                0 if tile_edge is None else int(tile_edge),
                                                float("nan") if lambda_ is None else float(lambda_), mode, ptrs, rsq.ctypes.data,
                                                C.byref(info)))
    return outs, rsq, {"n_slots": info.n_slots, "n_units": info.n_units, "step_ms": info.step_ms, "unit_ms_sum": info.unit_ms_sum,
                       "unit_ms_max": info.unit_ms_max, "slot_ms": [info.slot_ms[k] for k in range(info.n_slots)]}
