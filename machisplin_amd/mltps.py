"""Steps 2-5 of ``machisplin.mltps`` (V73:442-930) with every raster-sized operation on the
GPU: ensemble prediction over the covariate stack, thin-plate spline of the station
residuals (tiled exactly as the reference tiles it, or globally), seam feathering, the
final sum and the R^2 selection.  Model FITTING (Step 1 and the final fits) is out of scope:
callers pass the fitted members' parameters, as a `.Call()` shim would from R.
"""
from __future__ import annotations

import numpy as np

from . import _lib, tiles
from .models import ensemble_predict
from .raster import Geometry, RasterStack
from .tps import Tps, fit_many, interpolate


def station_predictors(stack: RasterStack, xy):
    """terra::extract(rast_stack, xy) (V73:145): covariates at the stations' cells plus the
    cell-centre LONG/LAT (the knots are CELL CENTRES, not the input coordinates)."""
    import torch
    g = stack.geom
    rows, cols = tiles.cells_from_xy(g, xy)
    inside = rows >= 0
    r = torch.from_numpy(np.where(inside, rows, 0)).to(stack.planes.device)
    c = torch.from_numpy(np.where(inside, cols, 0)).to(stack.planes.device)
    cov = stack.planes[:, r, c].to(torch.float64).cpu().numpy().T
    if not np.isnan(stack.nodata):
        cov[cov == stack.nodata] = np.nan
    cov[~inside] = np.nan
    X = np.column_stack([cov, g.x_from_col(cols), g.y_from_row(rows)])
    X[~inside, -2:] = np.nan
    return X, rows, cols


def ensemble_residuals(models, weights, wt_total, X, resp):
    """res.FINAL (V73:477-482 ... 608-611, 620): sum_k (resp - pred_k) * w_k / wt.tot."""
    import ctypes as C
    X = np.asfortranarray(np.asarray(X, dtype=np.float64))
    resp = np.ascontiguousarray(resp, dtype=np.float64)
    if len(models) > 8:      # the one-call entry point takes up to 8 members
        res = None
        for m, w in zip(models, weights):
            rk = (resp - m.predict_points(X)) * w
            res = rk if res is None else res + rk
        return res / wt_total
    hs = (C.c_void_p * len(models))(*[m._h for m in models])
    ws = (C.c_double * len(models))(*[float(w) for w in weights])
    out = np.empty(X.shape[0])
    _lib.check(_lib.lib().mhs_residual_points(hs, ws, len(models), float(wt_total), X.ctypes.data, resp.ctypes.data,
                                              X.shape[0], out.ctypes.data))
    return out


def tps_residual_surface(geom: Geometry, knots_xy, resid, cov1_at_stations=None, tile_edge: int = 1500,
                         lambda_=None, gcv_mode: str = "fields", out=None, info=None):
    """Step 3 + Step 4 (V73:636-897): thin-plate spline of the residuals over the whole grid.
    Grids larger than `tile_edge` in either direction are cut into ceil(n/tile_edge) tiles with
    their own fit on the stations of the +-20 % fit box, kept on the +-2.5 % box, mean-mosaicked
    and seam-feathered; otherwise one global fit (V73:748-753).  tile_edge=None forces the
    global fit (the north-star primitive).  Returns final.TPS as a device tensor."""
    import torch
    dev = torch.device("cuda", _lib.init())
    knots_xy = np.asarray(knots_xy, dtype=np.float64)
    resid = np.asarray(resid, dtype=np.float64)
    if info is None:
        # one call into the library (what the R shim binds): the tiles' fits run side by side on several lanes.
        # With `info` the same steps are composed here, tile by tile, and the per-tile station counts and
        # lambdas are reported (bit-identical result, tests/test_tiles_gpu.py).
        import ctypes as C
        if out is None:
            out = torch.empty((geom.nrow, geom.ncol), dtype=torch.float64, device=dev)
        if out.dtype != torch.float64 or not out.is_cuda or out.dim() != 2 or out.stride(1) != 1:
            raise ValueError("out must be a 2-D float64 device tensor with unit column stride")
        g = geom.c_struct()
        xyf = np.asfortranarray(knots_xy)
        cov = None if cov1_at_stations is None else np.ascontiguousarray(cov1_at_stations, dtype=np.float64)
        nt = (C.c_int64 * 2)()
        mode = {"fields": _lib.GCV_FIELDS, "converged": _lib.GCV_CONVERGED}[gcv_mode]
        st = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.lib().mhs_tps_surface_dev(C.byref(g), xyf.ctypes.data, resid.ctypes.data, resid.shape[0],
                                                  None if cov is None else cov.ctypes.data,
                                                  0 if tile_edge is None else int(tile_edge),
                                                  float("nan") if lambda_ is None else float(lambda_), mode,
                                                  out.data_ptr(), out.stride(0), nt, st))
        return out
    if tile_edge is None:
        nRx = nCx = 1
    else:
        nRx, nCx, fit_win, keep_win = tiles.step3_tile_windows(geom, tile_edge)
    if info is not None:
        info.update({"nRx": nRx, "nCx": nCx, "tile_n": [], "lambda": []})
    if nRx * nCx == 1:
        fit = Tps(knots_xy, resid, lambda_=lambda_, gcv_mode=gcv_mode)
        if info is not None:
            info["tile_n"].append(fit.n); info["lambda"].append(fit.lambda_)
        return interpolate(geom, fit, out=out)
    rows, cols = tiles.cells_from_xy(geom, knots_xy)
    ok = rows >= 0
    if cov1_at_stations is not None:
        ok &= ~np.isnan(np.asarray(cov1_at_stations, dtype=np.float64))
    # the tiles' stations first, then every tile's fields::Tps in ONE library call (mhs_tps_fit_many: a workgroup per
    # spline -- the arithmetic mhs_tps_surface itself uses for its tiles), then the evaluations
    sels = []
    for h in range(nRx * nCx):
        fr0, fr1, fc0, fc1 = (int(v) for v in fit_win[h])
        sels.append(np.flatnonzero(ok & (rows >= fr0) & (rows < fr1) & (cols >= fc0) & (cols < fc1)))
    todo = [h for h in range(nRx * nCx) if sels[h].size >= 10]
    fits = dict(zip(todo, fit_many([knots_xy[sels[h]] for h in todo], [resid[sels[h]] for h in todo],
                                   lambda_=lambda_, gcv_mode=gcv_mode)))
    bufs = []
    for h in range(nRx * nCx):
        fr0, fr1, fc0, fc1 = (int(v) for v in fit_win[h])
        kr0, kr1, kc0, kc1 = (int(v) for v in keep_win[h])
        if h not in fits:  # V73:710-721: the tile is all zeros
            bufs.append(torch.zeros((kr1 - kr0, kc1 - kc0), dtype=torch.float64, device=dev))
            if info is not None:
                info["tile_n"].append(int(sels[h].size)); info["lambda"].append(float("nan"))
            continue
        fit = fits[h]
        if fit is None:
            raise _lib.MhsError(_lib.ERR_NUMERIC, f"the spline of tile {h} could not be fitted")
        gf = geom.window(fr0, fr1, fc0, fc1)  # terra::rast(rb): geometry of the fit raster
        bufs.append(interpolate(gf, fit, window=(kr0 - fr0, kr1 - fr0, kc0 - fc0, kc1 - fc0)))
        if info is not None:
            info["tile_n"].append(fit.n); info["lambda"].append(fit.lambda_)
    return tiles.mosaic_feather(geom, nRx, nCx, keep_win, bufs, merge_mode=False, out=out)


def complete_cases(stack: RasterStack, int_values):
    """``Mydata[complete.cases(Mydata),]`` (V73:154): the reference filters ONCE over every column of the
    joined table -- long, lat, every response layer and every extracted covariate -- so a station with an NA
    in any response layer is dropped from ALL layers.  Returns the boolean keep mask over the rows of
    `int_values` (columns long, lat, response layers)."""
    int_values = np.asarray(int_values, dtype=np.float64)
    X, _, _ = station_predictors(stack, int_values[:, :2])
    return ~np.isnan(X).any(axis=1) & ~np.isnan(int_values).any(axis=1)


def mltps_predict(stack: RasterStack, int_xy, resp, models, weights, wt_total, tps: bool = True,
                  tile_edge: int = 1500, lambda_=None, gcv_mode: str = "fields", tps_info: bool = False,
                  keep=None):
    """machisplin.mltps Steps 2-5 for ONE response layer, given the fitted ensemble members.

    Returns a dict mirroring ``omega[[i]]`` (V73:914-930, 946-955): ``final`` (device tensor),
    ``residuals`` (n x 3: residual, long, lat), ``summary`` (r2 ensemble / r2 final) plus the
    intermediate ``pred_elev`` and ``final_tps`` planes.  ``tps_info=True`` composes Step 3 tile by tile and
    reports the per-tile station counts and lambdas in ``tps_info`` (same surface, slower).  ``keep`` is the
    table-wide complete.cases mask (:func:`complete_cases`, what :func:`mltps` passes); without it the mask is
    taken over this layer's columns only -- the caller then owns the reference's table-wide filter."""
    import torch
    g = stack.geom
    X, rows, cols = station_predictors(stack, int_xy)
    own = ~np.isnan(X).any(axis=1) & ~np.isnan(np.asarray(resp, dtype=np.float64))  # complete.cases, V73:154
    keep = own if keep is None else (np.asarray(keep, dtype=bool) & own)
    X, rows, cols = X[keep], rows[keep], cols[keep]
    y = np.asarray(resp, dtype=np.float64)[keep]
    # Step 2 (V73:447-620)
    pred_elev = ensemble_predict(stack, models, weights, wt_total)
    res_final = ensemble_residuals(models, weights, wt_total, X, y)
    tss = float(np.sum((y - y.mean()) ** 2))
    rsq_model = 1.0 - float(np.sum(res_final ** 2)) / tss
    out = {"pred_elev": pred_elev, "rsq_model": rsq_model, "n_stations": int(y.size)}
    knots = X[:, -2:]  # LONG, LAT columns of dat_tps (V73:688,751)
    if not tps:  # V73:934-953
        out.update({"final": pred_elev, "residuals": np.column_stack([res_final, knots]),
                    "summary": {"r2 ensemble": rsq_model}})
        return out
    # Step 3 + 4 (V73:636-897)
    info = {} if tps_info else None
    final_tps = tps_residual_surface(g, knots, res_final, cov1_at_stations=X[:, 0], tile_edge=tile_edge,
                                     lambda_=lambda_, gcv_mode=gcv_mode, info=info)
    # Step 5 (V73:902-930): sum, extract at the stations, keep the sum iff it improves R^2
    total = torch.empty_like(pred_elev)
    st = torch.cuda.current_stream(pred_elev.device).cuda_stream
    _lib.check(_lib.lib().mhs_scale_add_dev(pred_elev.data_ptr(), 1.0, final_tps.data_ptr(), total.data_ptr(),
                                            total.numel(), st))
    f_actual = tiles.extract(total, rows, cols)
    rsq_final = 1.0 - float(np.sum((y - f_actual) ** 2)) / tss
    out.update({"final_tps": final_tps, "rsq_final": rsq_final, "tps_info": info,
                "residuals": np.column_stack([y - f_actual, knots]),
                "summary": {"r2 ensemble": rsq_model, "r2 final": rsq_final},
                "final": total if rsq_final > rsq_model else pred_elev})
    return out


def mltps(stack: RasterStack, int_values, fitted, tps: bool = True, tile_edge: int = 1500, lambda_=None,
          gcv_mode: str = "fields"):
    """The layer loop of machisplin.mltps (V73:176-957) over fitted members: ``int_values`` is the reference's
    table as an array (columns long, lat, then one response column per layer, V73:120-154); ``fitted[i]`` holds
    layer i's ``models`` (device models in ``mods.run`` order), ``weights`` (rounded kept weights) and ``wt_total``
    (V73:337-392).  Returns the list ``omega``: one :func:`mltps_predict` result per layer plus ``n_layers``
    (V73:955) -- Step 1 (fitting, CV, weight search) happens before this call, in R or through :mod:`cv`."""
    int_values = np.asarray(int_values, dtype=np.float64)
    n_layers = int_values.shape[1] - 2
    if n_layers != len(fitted):
        raise ValueError("fitted must hold one entry per response column of int_values")
    keep = complete_cases(stack, int_values)   # once, over every column of the table (V73:154)
    omega = []
    from .tps import reduction_cache
    with reduction_cache():      # every layer fits the same stations: the tiles' reductions are built once (bit-identical fits)
        for i, f in enumerate(fitted):
            out = mltps_predict(stack, int_values[:, :2], int_values[:, 2 + i], f["models"], f["weights"], f["wt_total"],
                                tps=tps, tile_edge=tile_edge, lambda_=lambda_, gcv_mode=gcv_mode, keep=keep)
            out["n_layers"] = n_layers
            omega.append(out)
    return omega
