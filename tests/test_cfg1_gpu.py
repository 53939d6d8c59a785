"""BASELINE.json configs[0] on the GPU: the reference's bundled station table (sampling.csv, 813 stations,
bio_1 / bio_12) over the bundled TWI / slope rasters (level 0 of the .ovr files, INT2S, NoData -32768; a synthetic
ALT plane stands in for the missing alt.tif blob) through the whole Step 2-5 flow of machisplin.mltps, against the
oracle's literal restatement of V73:447-930.  At this geometry (1238 x 1632) the reference's tile rule (V73:656-661)
gives 1 x 2 Step-3 tiles, i.e. one vertical seam; the 2x nearest-neighbour upsampled grid (2476 x 3264, the full-
resolution geometry of the bundled rasters) gives the real 2 x 3 layout.  Data: tests/golden/cfg1_extdata.npz
(tests/golden/make_golden_cfg1.py)."""
import os

import numpy as np
import pytest

from oracle import cbind
from oracle import ensemble as oe
from oracle import tiles as ot
from oracle import tps as otps

pytestmark = pytest.mark.gpu

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg1_extdata.npz")
THREADS = min(16, os.cpu_count() or 1)


def _og(g):
    return ot.Geom(g.xmin, g.ymax, g.xres, g.yres, g.nrow, g.ncol)


def _cfg1(hip, upsample=1):
    import torch
    from machisplin_amd import synth
    z = np.load(FIX)
    xmin, ymax, xres, yres, nrow, ncol = z["geom"]
    g = hip.Geometry(float(xmin), float(ymax), float(xres) / upsample, float(yres) / upsample, int(nrow) * upsample,
                     int(ncol) * upsample)
    twi, slope = torch.from_numpy(z["TWI"]).cuda(), torch.from_numpy(z["slope"]).cuda()
    if upsample > 1:
        twi = twi.repeat_interleave(upsample, 0).repeat_interleave(upsample, 1)
        slope = slope.repeat_interleave(upsample, 0).repeat_interleave(upsample, 1)
    alt, _ = synth.covariates(g, 1, 813, dtype="i16")     # alt.tif is a missing blob: synthetic, alt-like range
    planes = torch.stack([alt[0], slope, twi]).contiguous()   # rast_stack order of the README example
    stack = hip.RasterStack(g, planes, float(z["nodata"]))
    tab = z["sampling"]
    host = planes.cpu().numpy().astype(np.float64)
    host[host == float(z["nodata"])] = np.nan
    return g, stack, host, tab


def _oracle_flow(g, host, X_grid_fn, xy, resp, params, wts, tot, tile_edge, lam):
    """Steps 2-5 literally (V73:447-930) with the TPS tiles fitted at the given lambdas; returns
    (pred, final_tps, final, rsq_model, rsq_final, n_stations, tile station counts)."""
    og = _og(g)
    rows = np.array([og.row_from_y(v) for v in xy[:, 1]])
    cols = np.array([og.col_from_x(v) for v in xy[:, 0]])
    x, y = otps.cell_centres(g.xmin, g.ymax, g.xres, g.yres, g.nrow, g.ncol)
    Xs = np.column_stack([host[:, rows, cols].T, x[cols], y[rows]])
    keep = ~np.isnan(Xs).any(axis=1) & ~np.isnan(resp)
    Xs, ys = Xs[keep], resp[keep]
    pred = X_grid_fn()
    res_final = None
    for p, w in zip(params, wts):
        rk = (ys - oe.predict(p, Xs)) * w
        res_final = rk if res_final is None else res_final + rk
    res_final = res_final / tot
    nRx, nCx, fw, kw = ot.step3_windows(og, tile_edge)
    knots = Xs[:, -2:]
    tiles, counts = [], []
    for h in range(nRx * nCx):
        sel = ot.stations_in_window(og, fw[h], knots, host[0])
        counts.append(int(sel.size))
        gf = ot.window_geom(og, fw[h])
        wk = (kw[h][0] - fw[h][0], kw[h][1] - fw[h][0], kw[h][2] - fw[h][2], kw[h][3] - fw[h][2])
        m = otps.fit(knots[sel], res_final[sel], lam=lam[h])
        tiles.append(cbind.tps_eval_grid(m, gf.xmin, gf.ymax, gf.xres, gf.yres, *wk, threads=THREADS))
    layers = [ot.extend_full(og, kw[h], tiles[h]) for h in range(len(kw))]
    final_tps = ot.feather_and_merge(og, nRx, nCx, kw, tiles, ot.mosaic_mean(layers[::-1]))
    final, rsq_model, rsq_final, resid = ot.step5_combine(og, pred, final_tps, knots, ys)
    return pred, final_tps, final, rsq_model, rsq_final, int(keep.sum()), counts, (nRx, nCx)


def test_cfg1_bundled_stations_one_by_two_tiles(hip):
    from machisplin_amd import synth
    g, stack, host, tab = _cfg1(hip)
    assert (g.nrow, g.ncol) == (1238, 1632) and tab.shape == (813, 4)
    xy, resp = tab[:, :2], tab[:, 2].copy()
    rows, cols = hip.tiles.cells_from_xy(g, xy)
    assert (rows >= 0).all()                                   # every bundled station falls on the grid
    x, y = otps.cell_centres(g.xmin, g.ymax, g.xres, g.yres, g.nrow, g.ncol)
    Xs = np.column_stack([host[:, rows, cols].T, x[cols], y[rows]])
    ok = ~np.isnan(Xs).any(axis=1)
    params = synth.ensemble_params(Xs[ok], resp[ok], 813, n_gbm_trees=400, n_rf_trees=25)
    kept, wts, tot = hip.models.select_weights(synth.OPTX_WEIGHTS)
    mods = [hip.models.from_param_dict(p) for p in params]
    res = hip.mltps_predict(stack, xy, resp, mods, wts, tot, tps=True, tile_edge=1500, tps_info=True)
    assert (res["tps_info"]["nRx"], res["tps_info"]["nCx"]) == (1, 2)       # ceil(1238/1500) x ceil(1632/1500), V73:656-661

    def grid_pred():
        Xg = oe.stack_predictors(host, (x, y))
        return cbind.ensemble(params, wts, tot, Xg, THREADS).reshape(g.nrow, g.ncol)

    pred, final_tps, final, rsq_model, rsq_final, n_st, counts, _ = _oracle_flow(
        g, host, grid_pred, xy, resp, params, wts, tot, 1500, res["tps_info"]["lambda"])
    assert res["n_stations"] == n_st and counts == res["tps_info"]["tile_n"]
    gp = res["pred_elev"].cpu().numpy()
    assert np.array_equal(np.isnan(gp), np.isnan(pred))
    assert np.nanmax(np.abs(gp - pred)) < 1e-10 * np.nanmax(np.abs(pred))
    gt = res["final_tps"].cpu().numpy()
    assert np.abs(gt - final_tps).max() < 1e-7 * np.abs(final_tps).max()
    assert abs(res["rsq_model"] - rsq_model) < 1e-9 and abs(res["rsq_final"] - rsq_final) < 1e-7
    got = res["final"].cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(final))
    assert np.nanmax(np.abs(got - final)) < 1e-7 * np.nanmax(np.abs(final))
    # the one-call default (tiles fitted side by side) gives the same planes bit for bit
    fast = hip.mltps_predict(stack, xy, resp, mods, wts, tot, tps=True, tile_edge=1500)
    assert np.array_equal(fast["final"].cpu().numpy(), got, equal_nan=True)


def test_cfg1_layer_loop_filters_complete_cases_once(hip):
    """machisplin.mltps over bio_1 and bio_12 with NAs in bio_12: complete.cases(Mydata) (V73:154) drops those
    stations from BOTH layers (mltps.py)."""
    from machisplin_amd import synth
    g, stack, host, tab = _cfg1(hip)
    iv = tab.copy()
    iv[[5, 77, 400], 3] = np.nan          # bio_12 missing at three stations
    keep = hip.mltps.complete_cases(stack, iv)
    assert keep.sum() <= 810 and not keep[[5, 77, 400]].any()
    rows, cols = hip.tiles.cells_from_xy(g, iv[:, :2])
    x, y = otps.cell_centres(g.xmin, g.ymax, g.xres, g.yres, g.nrow, g.ncol)
    Xs = np.column_stack([host[:, rows, cols].T, x[cols], y[rows]])
    fitted = []
    for k in range(2):
        params = synth.ensemble_params(Xs[keep], iv[keep, 2 + k], 900 + k, n_gbm_trees=60, n_rf_trees=5, which="gnmv")
        _, wts, tot = hip.models.select_weights([0.22, 0.12, 0.18, 0.41], labels="gnmv")
        fitted.append({"models": [hip.models.from_param_dict(p) for p in params], "weights": wts, "wt_total": tot})
    omega = hip.mltps.mltps(stack, iv, fitted, tile_edge=1500, lambda_=1e-3)
    assert len(omega) == 2 and omega[0]["n_layers"] == 2
    assert omega[0]["n_stations"] == omega[1]["n_stations"] == int(keep.sum())
    # layer 0 alone (its own column has no NA) would have kept three more stations
    alone = hip.mltps_predict(stack, iv[:, :2], iv[:, 2], fitted[0]["models"], fitted[0]["weights"], fitted[0]["wt_total"],
                              tile_edge=1500, lambda_=1e-3)
    assert alone["n_stations"] == int((~np.isnan(Xs).any(axis=1)).sum()) > omega[0]["n_stations"]


def test_cfg1_full_resolution_geometry_two_by_three_tiles(hip):
    """2476 x 3264 (the bundled rasters' own geometry): 2 x 3 Step-3 tiles, vertical and horizontal seams.  The
    ensemble plane is checked on sampled rows (the C oracle over all 8e6 cells of six members is minutes), the
    spline tiles, mosaic and feathering over the whole grid."""
    from machisplin_amd import synth
    g, stack, host, tab = _cfg1(hip, upsample=2)
    assert (g.nrow, g.ncol) == (2476, 3264)
    xy, resp = tab[:, :2], tab[:, 2].copy()
    rows, cols = hip.tiles.cells_from_xy(g, xy)
    x, y = otps.cell_centres(g.xmin, g.ymax, g.xres, g.yres, g.nrow, g.ncol)
    Xs = np.column_stack([host[:, rows, cols].T, x[cols], y[rows]])
    ok = ~np.isnan(Xs).any(axis=1)
    params = synth.ensemble_params(Xs[ok], resp[ok], 814, n_gbm_trees=300, n_rf_trees=20)
    kept, wts, tot = hip.models.select_weights(synth.OPTX_WEIGHTS)
    mods = [hip.models.from_param_dict(p) for p in params]
    res = hip.mltps_predict(stack, xy, resp, mods, wts, tot, tps=True, tile_edge=1500, tps_info=True)
    assert (res["tps_info"]["nRx"], res["tps_info"]["nCx"]) == (2, 3)
    gp = res["pred_elev"].cpu().numpy()
    sample = np.array([0, 1, 617, 1237, 1238, 1900, 2475])
    for r in sample:
        Xg = oe.stack_predictors(host[:, r:r + 1], (x, y[r:r + 1]))
        want = cbind.ensemble(params, wts, tot, Xg, THREADS)
        assert np.array_equal(np.isnan(gp[r]), np.isnan(want))
        assert np.nanmax(np.abs(gp[r] - want)) < 1e-10 * np.nanmax(np.abs(want))
    # Steps 3-5 over the whole grid with the GPU's ensemble plane as Step 2 (checked above on the sample)
    pred, final_tps, final, rsq_model, rsq_final, n_st, counts, lay = _oracle_flow(
        g, host, lambda: gp, xy, resp, params, wts, tot, 1500, res["tps_info"]["lambda"])
    assert lay == (2, 3) and counts == res["tps_info"]["tile_n"] and res["n_stations"] == n_st
    gt = res["final_tps"].cpu().numpy()
    assert np.abs(gt - final_tps).max() < 1e-7 * np.abs(final_tps).max()
    assert abs(res["rsq_final"] - rsq_final) < 1e-7
    got = res["final"].cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(final))
    assert np.nanmax(np.abs(got - final)) < 1e-7 * np.nanmax(np.abs(final))
