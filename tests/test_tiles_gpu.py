"""GPU parity: mosaic / feather / overlay kernels and the tiled Step 2-5 flow (through the
C ABI) vs the oracle's literal restatement of V73:636-930 and tiles.create/merge."""
import numpy as np
import pytest

from oracle import ensemble as oe
from oracle import tiles as ot
from oracle import tps as otps

pytestmark = pytest.mark.gpu


def _og(g):
    return ot.Geom(g.xmin, g.ymax, g.xres, g.yres, g.nrow, g.ncol)


def _rand_tiles(windows, seed, nan_frac=0.0):
    rng = np.random.default_rng(seed)
    out = []
    for w in windows:
        t = rng.standard_normal((w[1] - w[0], w[3] - w[2]))
        if nan_frac:
            t[rng.random(t.shape) < nan_frac] = np.nan
        out.append(t)
    return out


@pytest.mark.parametrize("nrow,ncol,edge,xmin,ymax", [(300, 400, 100, -78.0, -5.0), (130, 250, 128, 10.25, 45.5),
                                                      (260, 120, 128, -0.05, 0.11), (90, 95, 100, -78.0, -5.0)])
def test_step4_mosaic_feather_matches_oracle(hip, nrow, ncol, edge, xmin, ymax):
    import torch
    g = hip.Geometry(xmin, ymax, 1 / 1200, 1 / 1200, nrow, ncol)
    nRx, nCx, fit, keep = hip.tiles.step3_tile_windows(g, edge)
    tiles = _rand_tiles(keep, nrow)
    dev = [torch.from_numpy(t).cuda() for t in tiles]
    got, seams = hip.tiles.mosaic_feather(g, nRx, nCx, keep, dev, return_seams=True)
    og = _og(g)
    kw = [tuple(int(v) for v in w) for w in keep]
    layers = [ot.extend_full(og, kw[h], tiles[h]) for h in range(len(kw))]
    base = ot.mosaic_mean(layers[::-1])
    want = ot.feather_and_merge(og, nRx, nCx, kw, tiles, base)
    assert np.array_equal(got.cpu().numpy(), want, equal_nan=True)  # same operations, same order: same bits


def test_tiles_merge_with_na_cells_matches_oracle(hip):
    import torch
    g = hip.Geometry(-78.0, -5.0, 1 / 1200, 1 / 1200, 240, 330)
    xy = np.random.default_rng(0).uniform([g.xmin, g.ymin], [g.xmax, g.ymax], (50, 2))
    t = hip.tiles.tiles_create(g, xy, out_ncol=3, out_nrow=2, feather_d=30)
    tiles = _rand_tiles(t["win"], 3, nan_frac=0.05)
    # a NA band at a tile edge shrinks the as.points() bounding box of the seam
    tiles[0][:, -4:] = np.nan
    dev = [torch.from_numpy(x).cuda() for x in tiles]
    got = hip.tiles.tiles_merge(g, t["win"], dev, in_ncol=3, in_nrow=2).cpu().numpy()
    wins = [tuple(int(v) for v in w) for w in t["win"]]
    want = ot.tiles_merge(_og(g), wins, tiles, 3, 2)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.allclose(got, want, rtol=1e-15, atol=0, equal_nan=True)
    # host-pointer entry point (what the R shim binds for machisplin.tiles.merge)
    import ctypes as C
    from machisplin_amd import _lib
    host_tiles = [np.ascontiguousarray(x) for x in tiles]
    ptrs = (C.c_void_p * 6)(*[x.ctypes.data for x in host_tiles])
    out = np.empty((g.nrow, g.ncol))
    gs = g.c_struct()
    win = np.ascontiguousarray(t["win"], dtype=np.int64)
    _lib.check(_lib.lib().mhs_mosaic_feather(C.byref(gs), 2, 3, win.ctypes.data, ptrs, 1, out.ctypes.data))
    assert np.array_equal(out, got, equal_nan=True)


def _ensemble_inputs(hip, nrow, ncol, n, seed):
    from machisplin_amd import synth
    g = synth.grid(nrow, ncol)
    planes, nodata = synth.covariates(g, 3, seed, dtype="f32", nodata_frac=0.002)
    stack = hip.RasterStack(g, planes, nodata)
    xy, rows, cols, uv = synth.stations(g, n, seed)
    host = planes.cpu().numpy().astype(np.float64)
    x, y = otps.cell_centres(g.xmin, g.ymax, g.xres, g.yres, nrow, ncol)
    X = oe.stack_predictors(host, (x, y))
    Xs = X[rows * ncol + cols]
    resp = synth.response(np.nan_to_num(Xs), uv, seed)
    ok = ~np.isnan(Xs).any(axis=1)
    params = synth.ensemble_params(Xs[ok], resp[ok], seed, n_gbm_trees=150, n_rf_trees=10)
    return g, stack, host, X, xy, resp, params


@pytest.mark.parametrize("tile_edge", [100, 1500])
def test_mltps_steps_2_to_5_end_to_end(hip, tile_edge):
    """G8 mini: 600 stations, 3 covariates, 220 x 290 grid; tile edge 100 forces 3 x 3 TPS tiles
    (edge 1500 takes the single-tile branch V73:748-753)."""
    g, stack, host, X, xy, resp, params = _ensemble_inputs(hip, 220, 290, 600, 31)
    kept, wts, tot = hip.models.select_weights([0.31, 0.22, 0.12, 0.18, 0.27, 0.41])
    mods = [hip.models.from_param_dict(p) for p in params]
    res = hip.mltps_predict(stack, xy, resp, mods, wts, tot, tps=True, tile_edge=tile_edge, tps_info=True)
    # the default takes the one-call Step 3 + 4 (tiles fitted side by side): same planes, bit for bit
    fast = hip.mltps_predict(stack, xy, resp, mods, wts, tot, tps=True, tile_edge=tile_edge)
    assert "tps_info" in fast and fast["tps_info"] is None
    assert np.array_equal(fast["final"].cpu().numpy(), res["final"].cpu().numpy(), equal_nan=True)
    assert fast["rsq_final"] == res["rsq_final"]
    # the layer loop of machisplin.mltps over two response columns (the second one shifted)
    iv = np.column_stack([xy, resp, resp + 3.0])
    omega = hip.mltps.mltps(stack, iv, [{"models": mods, "weights": wts, "wt_total": tot}] * 2, tile_edge=tile_edge)
    assert len(omega) == 2 and omega[0]["n_layers"] == 2
    assert np.array_equal(omega[0]["final"].cpu().numpy(), res["final"].cpu().numpy(), equal_nan=True)
    assert omega[1]["rsq_model"] < omega[0]["rsq_model"]      # same members, response shifted by 3: a worse ensemble fit
    # ---- oracle: the same flow, literally (V73:447-930)
    og = _og(g)
    rows = np.array([og.row_from_y(v) for v in xy[:, 1]])
    cols = np.array([og.col_from_x(v) for v in xy[:, 0]])
    Xs = X[rows * g.ncol + cols]
    keep = ~np.isnan(Xs).any(axis=1)
    Xs, y = Xs[keep], resp[keep]
    pred = oe.ensemble(params, wts, tot, X).reshape(g.nrow, g.ncol)
    res_final = None
    for p, w in zip(params, wts):
        rk = (y - oe.predict(p, Xs)) * w
        res_final = rk if res_final is None else res_final + rk
    res_final = res_final / tot
    assert res["n_stations"] == keep.sum()
    assert np.nanmax(np.abs(res["pred_elev"].cpu().numpy() - pred)) < 1e-10 * np.nanmax(np.abs(pred))
    # TPS at the GPU's own lambdas (flat GCV minimum), then the bookkeeping must agree exactly
    lam = res["tps_info"]["lambda"]
    nRx, nCx, fw, kw = ot.step3_windows(og, tile_edge)
    assert (nRx, nCx) == (res["tps_info"]["nRx"], res["tps_info"]["nCx"])
    knots = Xs[:, -2:]
    if nRx * nCx == 1:
        final_tps = otps.predict_grid(otps.fit(knots, res_final, lam=lam[0]), og.xmin, og.ymax, og.xres, og.yres, og.nrow, og.ncol)
    else:
        tiles = []
        for h in range(nRx * nCx):
            sel = ot.stations_in_window(og, fw[h], knots, host[0])
            assert sel.size == res["tps_info"]["tile_n"][h]
            gf = ot.window_geom(og, fw[h])
            wk = (kw[h][0] - fw[h][0], kw[h][1] - fw[h][0], kw[h][2] - fw[h][2], kw[h][3] - fw[h][2])
            m = otps.fit(knots[sel], res_final[sel], lam=lam[h])
            tiles.append(otps.predict_grid(m, gf.xmin, gf.ymax, gf.xres, gf.yres, gf.nrow, gf.ncol, *wk))
        layers = [ot.extend_full(og, kw[h], tiles[h]) for h in range(len(kw))]
        final_tps = ot.feather_and_merge(og, nRx, nCx, kw, tiles, ot.mosaic_mean(layers[::-1]))
    got_tps = res["final_tps"].cpu().numpy()
    assert np.abs(got_tps - final_tps).max() < 1e-7 * np.abs(final_tps).max()
    final, rsq_model, rsq_final, resid = ot.step5_combine(og, pred, final_tps, knots, y)
    assert abs(res["rsq_model"] - rsq_model) < 1e-9 and abs(res["rsq_final"] - rsq_final) < 1e-7
    got = res["final"].cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(final))
    assert np.nanmax(np.abs(got - final)) < 1e-7 * np.nanmax(np.abs(final))
    assert np.abs(res["residuals"][:, 0] - resid).max() < 1e-6
    # and with the oracle's own GCV lambdas the surface stays within the north-star 1e-6
    if nRx * nCx == 1:
        ref = otps.predict_grid(otps.fit(knots, res_final), og.xmin, og.ymax, og.xres, og.yres, og.nrow, og.ncol)
        assert np.abs(got_tps - ref).max() < 1e-6 * np.abs(ref).max()


def test_sparse_tile_becomes_zero_tile(hip):
    """V73:710-721: fewer than 10 stations in a fit box => that tile is all zeros."""
    g = hip.Geometry(-78.0, -5.0, 1 / 1200, 1 / 1200, 200, 200)
    rng = np.random.default_rng(2)
    # stations only in the western half
    cols = rng.integers(0, 60, 300)
    rows = rng.integers(0, 200, 300)
    cells = np.unique(rows * 200 + cols)
    rows, cols = np.divmod(cells, 200)
    xy = np.column_stack([g.x_from_col(cols), g.y_from_row(rows)])
    resid = rng.standard_normal(xy.shape[0])
    info = {}
    surf = hip.tps_residual_surface(g, xy, resid, tile_edge=100, lambda_=1e-2, info=info).cpu().numpy()
    assert info["nRx"] == 2 and info["nCx"] == 2
    assert min(info["tile_n"]) < 10 and max(info["tile_n"]) > 100
    nRx, nCx, fit, keep = hip.tiles.step3_tile_windows(g, 100)
    east_only = np.zeros((200, 200), dtype=bool)
    for h in (1, 3):
        east_only[keep[h][0]:keep[h][1], keep[h][2]:keep[h][3]] = True
    for h in (0, 2):
        east_only[keep[h][0]:keep[h][1], keep[h][2]:keep[h][3]] = False
    assert (surf[east_only] == 0).all() and not np.isnan(surf).any()


def test_extract_gathers_station_cells(hip):
    import torch
    g = hip.Geometry(-78.0, -5.0, 0.01, 0.01, 50, 60)
    plane = torch.arange(50 * 60, dtype=torch.float64, device="cuda").reshape(50, 60)
    xy = np.array([[-77.995, -5.005], [-77.405, -5.495], [-79.0, -5.1]])
    rows, cols = hip.tiles.cells_from_xy(g, xy)
    got = hip.tiles.extract(plane, rows, cols)
    assert got[0] == 0 and got[1] == 49 * 60 + 59 and np.isnan(got[2])


def test_one_call_surface_equals_composed_steps(hip):
    """mhs_tps_surface (what the .Call() shim binds for V73:636-897) == the Python composition of
    tile windows + per-tile Tps + interpolate + mosaic_feather, bit for bit."""
    import ctypes as C
    from machisplin_amd import _lib, synth
    g = synth.grid(230, 310)
    xy, rows, cols, uv = synth.stations(g, 500, 77)
    resid = synth.tps_residual(uv, 77)
    cov1 = np.ones(500)
    cov1[::37] = np.nan  # stations on NA covariate cells are dropped (V73:701-706)
    info = {}   # with `info` the Python mirror composes the steps itself, tile by tile
    want = hip.tps_residual_surface(g, xy, resid, cov1_at_stations=cov1, tile_edge=100, info=info).cpu().numpy()
    assert (info["nRx"], info["nCx"]) == (3, 4) and len(info["tile_n"]) == 12
    # without it, the mirror makes the one library call (device output)
    assert np.array_equal(hip.tps_residual_surface(g, xy, resid, cov1_at_stations=cov1, tile_edge=100).cpu().numpy(), want)
    out = np.empty((230, 310))
    nt = np.zeros(2, dtype=np.int64)
    gs = g.c_struct()
    xyf = np.asfortranarray(xy)
    _lib.check(_lib.lib().mhs_tps_surface(C.byref(gs), xyf.ctypes.data, resid.ctypes.data, 500, cov1.ctypes.data, 100,
                                          float("nan"), 0, out.ctypes.data, nt.ctypes.data))
    assert list(nt) == [3, 4]
    assert np.array_equal(out, want)
    # single-tile branch
    _lib.check(_lib.lib().mhs_tps_surface(C.byref(gs), xyf.ctypes.data, resid.ctypes.data, 500, None, 0,
                                          float("nan"), 0, out.ctypes.data, nt.ctypes.data))
    assert list(nt) == [1, 1]
    assert np.array_equal(out, hip.tps_residual_surface(g, xy, resid, tile_edge=None, info={}).cpu().numpy())


def test_cfg4_chain_tiles_create_per_tile_mltps_tiles_merge(hip):
    """BASELINE config 4 in miniature: machisplin.tiles.create (2 x 2, feather.d) -> an independent
    mltps run per tile and response layer with the smooth members only (g, n, m, v -- V73:366-392)
    -> machisplin.tiles.merge.  Oracle: the same chain, literally."""
    import torch
    from machisplin_amd import synth
    g, stack, host, X, xy, resp, params = _ensemble_inputs(hip, 260, 340, 900, 43)
    og = _og(g)
    smooth = [params[1], params[2], params[3], params[5]]  # g, n, m, v
    kept, wts, tot = hip.models.select_weights([0.22, 0.12, 0.18, 0.41], labels="gnmv")
    assert kept == "gnmv"
    t = hip.tiles.tiles_create(g, xy, out_ncol=2, out_nrow=2, feather_d=30)
    boxes, owins, osel = ot.tiles_create(og, xy, 2, 2, 30)
    assert np.array_equal(t["win"], np.array(owins))
    for layer in range(2):  # two response layers
        y = resp + 0.5 * layer * np.sin(3 * xy[:, 0])
        finals_gpu, finals_cpu = [], []
        for h in range(4):
            r0, r1, c0, c1 = (int(v) for v in t["win"][h])
            sel = t["dat"][h]
            assert np.array_equal(sel, osel[h])
            sub = hip.RasterStack(t["geom"][h], stack.planes[:, r0:r1, c0:c1].contiguous(), stack.nodata)
            mods = [hip.models.from_param_dict(p) for p in smooth]
            res = hip.mltps_predict(sub, xy[sel], y[sel], mods, wts, tot, tps=True, tile_edge=100, tps_info=True)
            finals_gpu.append(res["final"].contiguous())
            # ---- oracle for this tile
            tg = ot.window_geom(og, (r0, r1, c0, c1))
            Xt = oe.stack_predictors(host[:, r0:r1, c0:c1], otps.cell_centres(tg.xmin, tg.ymax, tg.xres, tg.yres, tg.nrow, tg.ncol))
            rows = np.array([tg.row_from_y(v) for v in xy[sel, 1]])
            cols = np.array([tg.col_from_x(v) for v in xy[sel, 0]])
            Xs = Xt[rows * tg.ncol + cols]
            keep = ~np.isnan(Xs).any(axis=1)
            Xs, ys = Xs[keep], y[sel][keep]
            pred = oe.ensemble(smooth, wts, tot, Xt).reshape(tg.nrow, tg.ncol)
            rf = None
            for p, w in zip(smooth, wts):
                rk = (ys - oe.predict(p, Xs)) * w
                rf = rk if rf is None else rf + rk
            rf = rf / tot
            lam = res["tps_info"]["lambda"]
            nRx, nCx, fw, kw = ot.step3_windows(tg, 100)
            tl = []
            for k in range(nRx * nCx):
                s2 = ot.stations_in_window(tg, fw[k], Xs[:, -2:], host[0, r0:r1, c0:c1])
                gf = ot.window_geom(tg, fw[k])
                wk = (kw[k][0] - fw[k][0], kw[k][1] - fw[k][0], kw[k][2] - fw[k][2], kw[k][3] - fw[k][2])
                if s2.size < 10:
                    tl.append(np.zeros((kw[k][1] - kw[k][0], kw[k][3] - kw[k][2])))
                else:
                    tl.append(otps.predict_grid(otps.fit(Xs[s2, -2:], rf[s2], lam=lam[k]), gf.xmin, gf.ymax, gf.xres, gf.yres,
                                                gf.nrow, gf.ncol, *wk))
            layers = [ot.extend_full(tg, kw[k], tl[k]) for k in range(len(kw))]
            ftps = ot.feather_and_merge(tg, nRx, nCx, kw, tl, ot.mosaic_mean(layers[::-1]))
            fin, rm, rfin, _ = ot.step5_combine(tg, pred, ftps, Xs[:, -2:], ys)
            assert abs(res["rsq_final"] - rfin) < 1e-6
            finals_cpu.append(fin)
        merged = hip.tiles.tiles_merge(g, t["win"], finals_gpu, in_ncol=2, in_nrow=2).cpu().numpy()
        want = ot.tiles_merge(og, owins, finals_cpu, 2, 2)
        assert np.array_equal(np.isnan(merged), np.isnan(want))
        assert np.nanmax(np.abs(merged - want)) < 1e-7 * np.nanmax(np.abs(want))
