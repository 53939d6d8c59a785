"""Host-side mirror of the reference's TPS interface for the hot path.

``Tps(x, Y)``            <-> ``fields::Tps(x, Y)``                    V73:722, V73:751
``interpolate(geom, m)`` <-> ``terra::interpolate(terra::rast(rb), m)`` V73:726, V73:753
``m.predict(xy)``        <-> ``predict(m, xy)`` (predict.Krig)

All arithmetic runs in libmachisplin_hip.so (HIP, gfx950); this module only marshals
arguments.  There is no CPU path.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import math

import numpy as np

from . import _lib
from .raster import Geometry


class Tps:
    """Fitted thin-plate smoothing spline, the subset of fields' ``Krig`` object that
    ``predict.Krig`` reads: ``c``, ``d``, ``lambda_``, ``knots`` (range-scaled),
    ``center``/``scale`` (``$transform$x.center`` / ``$x.scale``), ``eff_df``, ``gcv``."""

    def __init__(self, x, Y, lambda_: float | None = None, gcv_mode: str = "fields"):
        x = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
        if x.ndim != 2 or x.shape[1] != 2:
            raise ValueError("x must be an N x 2 matrix of (LONG, LAT)")
        y = np.ascontiguousarray(np.asarray(Y, dtype=np.float64).reshape(-1))
        if y.shape[0] != x.shape[0]:
            raise ValueError("x and Y have different numbers of rows")
        xy_cm = np.asfortranarray(x)  # column-major, as R hands it over
        mode = {"fields": _lib.GCV_FIELDS, "converged": _lib.GCV_CONVERGED}[gcv_mode]
        lam = math.nan if lambda_ is None else float(lambda_)
        h = C.c_void_p()
        _lib.check(_lib.lib().mhs_tps_fit(xy_cm.ctypes.data, y.ctypes.data, x.shape[0], lam, mode,
                                          C.byref(h)))
        self._h = h
        self._pull()

    @classmethod
    def from_coef(cls, knots_uv, c, d, lambda_, center, scale) -> "Tps":
        """Wrap coefficients captured elsewhere (e.g. a real fields::Tps object)."""
        self = cls.__new__(cls)
        kn = np.asfortranarray(np.asarray(knots_uv, dtype=np.float64))
        c = np.ascontiguousarray(np.asarray(c, dtype=np.float64))
        d = np.ascontiguousarray(np.asarray(d, dtype=np.float64))
        ce = np.ascontiguousarray(np.asarray(center, dtype=np.float64))
        sc = np.ascontiguousarray(np.asarray(scale, dtype=np.float64))
        if kn.ndim != 2 or kn.shape[1] != 2 or c.shape != (kn.shape[0],) or d.shape != (3,):
            raise ValueError("knots must be n x 2, c length n, d length 3")
        h = C.c_void_p()
        _lib.check(_lib.lib().mhs_tps_from_coef(kn.ctypes.data, c.ctypes.data, d.ctypes.data,
                                                kn.shape[0], float(lambda_), ce.ctypes.data,
                                                sc.ctypes.data, C.byref(h)))
        self._h = h
        self._pull()
        return self

    def _pull(self):
        lib = _lib.lib()
        n = C.c_int64()
        _lib.check(lib.mhs_tps_size(self._h, C.byref(n)))
        n = n.value
        self.c = np.empty(n)
        self.d = np.empty(3)
        kn = np.empty((n, 2), order="F")
        lam, edf, gcv = C.c_double(), C.c_double(), C.c_double()
        self.center, self.scale = np.empty(2), np.empty(2)
        _lib.check(lib.mhs_tps_get(self._h, self.c.ctypes.data, self.d.ctypes.data, kn.ctypes.data,
                                   C.byref(lam), self.center.ctypes.data, self.scale.ctypes.data,
                                   C.byref(edf), C.byref(gcv)))
        self.knots = np.ascontiguousarray(kn)
        self.lambda_, self.eff_df, self.gcv = lam.value, edf.value, gcv.value
        self.n = n

    def predict(self, xy) -> np.ndarray:
        """predict(tps, xy) at arbitrary points (n x 2: LONG, LAT)."""
        xy = np.asfortranarray(np.asarray(xy, dtype=np.float64).reshape(-1, 2))
        out = np.empty(xy.shape[0])
        _lib.check(_lib.lib().mhs_tps_predict_points(self._h, xy.ctypes.data, xy.shape[0],
                                                     out.ctypes.data))
        return out

    def eval_plan(self):
        """(tile_cols, tile_rows, node_pairs, cell_pairs) of this handle's last grid evaluation; tile 0 x 0 means
        the direct sum ran."""
        tc, tr, a, b = C.c_int(0), C.c_int(0), C.c_int64(0), C.c_int64(0)
        _lib.check(_lib.lib().mhs_tps_eval_plan(self._h, C.byref(tc), C.byref(tr), C.byref(a), C.byref(b)))
        return tc.value, tr.value, a.value, b.value

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and _lib._lib is not None:
            _lib._lib.mhs_tps_free(h)
            self._h = None


def fit_many(xs, Ys, lambda_: float | None = None, gcv_mode: str = "fields"):
    """``[fields::Tps(x, Y) for x, Y in zip(xs, Ys)]`` in ONE library call (mhs_tps_fit_many): every fit with 8..256
    distinct locations -- the tiles of the reference's Step 3, V73:690-738 -- is done by a single kernel launch, one
    workgroup per fit.  Returns a list of :class:`Tps` (None where a fit failed, e.g. collinear stations)."""
    lib = _lib.lib()
    k = len(xs)
    if len(Ys) != k:
        raise ValueError("xs and Ys have different lengths")
    xcm, ycm = [], []
    for x, Y in zip(xs, Ys):
        x = np.asarray(x, dtype=np.float64)
        if x.ndim != 2 or x.shape[1] != 2:
            raise ValueError("every x must be an N x 2 matrix of (LONG, LAT)")
        y = np.ascontiguousarray(np.asarray(Y, dtype=np.float64).reshape(-1))
        if y.shape[0] != x.shape[0]:
            raise ValueError("x and Y have different numbers of rows")
        xcm.append(np.asfortranarray(x)); ycm.append(y)
    _lib.init()
    px = (C.c_void_p * k)(*[a.ctypes.data for a in xcm])
    py = (C.c_void_p * k)(*[a.ctypes.data for a in ycm])
    ns = (C.c_int64 * k)(*[a.shape[0] for a in xcm])
    out = (C.c_void_p * k)()
    st = (C.c_int * k)()
    mode = {"fields": _lib.GCV_FIELDS, "converged": _lib.GCV_CONVERGED}[gcv_mode]
    lam = math.nan if lambda_ is None else float(lambda_)
    _lib.check(lib.mhs_tps_fit_many(px, py, ns, k, lam, mode, out, st))
    fits = []
    for i in range(k):
        if not out[i]:
            fits.append(None)
            continue
        t = Tps.__new__(Tps)
        t._h = C.c_void_p(out[i])
        t._pull()
        fits.append(t)
    return fits


def interpolate(geom: Geometry, model: Tps, window=None, out=None, stream=None, rows=None):
    """terra::interpolate(geometry-only raster, tps): evaluate the spline at EVERY cell
    centre of ``geom`` (or of the window (r0, r1, c0, c1)); no NA mask (V73:726,753).

    Returns a float64 torch tensor on the GPU, shape (r1-r0, c1-c0), row-major from the
    north-west cell.  ``out`` may be a pre-allocated 2-D device tensor (row stride = ld).
    ``rows=(b0, b1)``: only those rows of the window, evaluated with the WINDOW's plan (a device's row band: the bands
    of several devices stitch to the one-piece plane bit for bit); ``out`` then has b1 - b0 rows.
    """
    import torch
    r0, r1, c0, c1 = window if window is not None else (0, geom.nrow, 0, geom.ncol)
    b0, b1 = rows if rows is not None else (r0, r1)
    dev = torch.device("cuda", _lib.init())
    if out is None:
        out = torch.empty((b1 - b0, c1 - c0), dtype=torch.float64, device=dev)
    if out.dtype != torch.float64 or not out.is_cuda or out.dim() != 2 or out.stride(1) != 1:
        raise ValueError("out must be a 2-D float64 device tensor with unit column stride")
    if tuple(out.shape) != (b1 - b0, c1 - c0):
        raise ValueError("out has the wrong shape for the window")
    g = geom.c_struct()
    s = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
    if b1 > b0:
        _lib.check(_lib.lib().mhs_tps_predict_rows_dev(model._h, C.byref(g), r0, r1, c0, c1, b0, b1,
                                                       out.data_ptr(), out.stride(0), s))
    return out


EVAL_AUTO, EVAL_DIRECT, EVAL_FAR_FIELD = 0, 1, 2


@contextlib.contextmanager
def reduction_cache():
    """Scope of mhs_tps_reduction_cache: inside it, small GCV fits (the reference-tiled mode's tiles) of a station set
    that has been fitted before -- another response layer of the same table -- reuse its reduction and send only
    their right-hand side through it; same coefficients bit for bit.  Everything kept is freed on exit."""
    _lib.init()
    _lib.check(_lib.lib().mhs_tps_reduction_cache(1))
    try:
        yield
    finally:
        _lib.check(_lib.lib().mhs_tps_reduction_cache(0))


def eval_mode(mode: int) -> None:
    """How :func:`interpolate` sums the knots: EVAL_AUTO (by cost), EVAL_DIRECT (every knot for
    every cell, predict.Krig's own arithmetic) or EVAL_FAR_FIELD (direct sum over the knots near a
    tile + 16 x 16 Chebyshev interpolation of the analytic sum over the rest; equal to the direct sum
    to FP64 rounding)."""
    _lib.init()
    _lib.check(_lib.lib().mhs_tps_eval_mode(int(mode)))
