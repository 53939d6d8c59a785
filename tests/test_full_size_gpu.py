"""GPU, at BASELINE.json's FULL sizes, through size-independent properties (no oracle run at
these sizes): two independent HIP implementations of the same member must agree bit for bit,
windows must tile the full-grid result exactly, the fitted spline must satisfy its normal
equations, and sampled rows must match the C oracle."""
import os

import numpy as np
import pytest

from oracle import cbind, ensemble as oe, tps as otps

pytestmark = pytest.mark.gpu


def _cfg3_stack(hip, side, dtype="f32"):
    from machisplin_amd import synth
    g = synth.grid(side, side)
    planes, nodata = synth.covariates(g, 3, synth.BASE_SEED + 3, dtype=dtype, nodata_frac=0.001)
    return g, hip.RasterStack(g, planes, nodata), planes


def test_cfg3_tree_members_fast_path_equals_generic_walk_on_1e8_cells(hip, monkeypatch):
    """gbm: predicate-LUT kernel vs node walk; randomForest: level-synchronous LDS walk vs node walk.
    float32 planes take the fast kernels with float keys, the same values as float64 planes (what the R shim
    hands over) take them with double keys, and MHS_TREES_GENERIC=1 forces the node walk; the forest kernels and gbm's
    tree-order kernels sum the trees in the walk's order, so their 10 000 x 10 000 planes must be IDENTICAL to it (gbm's
    default coherent kernel: to rounding)."""
    import torch
    from machisplin_amd import synth
    side = 10000
    g, stack32, planes = _cfg3_stack(hip, side)
    xy, rows, cols, uv = synth.stations(g, 5000, synth.BASE_SEED + 3)
    cov = planes[:, torch.from_numpy(rows).cuda(), torch.from_numpy(cols).cuda()].cpu().numpy().astype(np.float64).T
    X = np.column_stack([np.nan_to_num(cov), xy])
    y = synth.response(X, uv, 1)
    stack64 = hip.RasterStack(g, planes.to(torch.float64), float("nan"))  # same values as doubles
    for prm in (synth.gbm_params(X, y, 3, n_trees=10000), synth.rf_params(X, y, 3, n_trees=500)):
        m = hip.models.from_param_dict(prm)
        a = hip.predict(stack32, m)
        b = hip.predict(stack64, m)
        monkeypatch.setenv("MHS_TREES_GENERIC", "1")
        c = hip.predict(stack64, m)
        monkeypatch.delenv("MHS_TREES_GENERIC")
        # float32 and float64 planes of the same values: the same ranks, the same kernel, the same bits
        assert torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b)), prm["kind"]
        assert torch.equal(torch.isnan(a), torch.isnan(c)), prm["kind"]
        if prm["kind"] == "gbm":
            # the default for grids, gbm_coherent_kernel, sums a cell's trees in another order than the walk: to rounding;
            # the tree-order kernels (MHS_GBM_NO_COHERENT=1) are the walk bit for bit
            assert float((torch.nan_to_num(a) - torch.nan_to_num(c)).abs().max()) <= 1e-13 * float(torch.nan_to_num(c).abs().max())
            monkeypatch.setenv("MHS_GBM_NO_COHERENT", "1")
            a = hip.predict(stack32, m)
            monkeypatch.delenv("MHS_GBM_NO_COHERENT")
        assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(c)), prm["kind"]
        del a, b, c
    del stack64
    # sampled rows against the C oracle (gbm NA routing included)
    host = planes[:, 4321:4323].cpu().numpy().astype(np.float64)
    xs, ys = otps.cell_centres(g.xmin, g.ymax, g.xres, g.yres, side, side, 4321, 4323)
    Xg = oe.stack_predictors(host, (xs, ys))
    prm = synth.rf_params(X, y, 3, n_trees=500)
    got = hip.predict(stack32, hip.models.from_param_dict(prm), window=(4321, 4323, 0, side)).cpu().numpy().ravel()
    want = cbind.predict(prm, Xg, threads=8)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.nanmax(np.abs(got - want)) < 1e-11 * np.nanmax(np.abs(want))


def test_cfg2_tps_windows_tile_the_full_grid_and_match_c_oracle(hip):
    import torch
    from machisplin_amd import synth
    g = synth.grid(2000, 2000)
    xy, rows, cols, uv = synth.stations(g, 2000, synth.BASE_SEED + 2)
    fit = hip.Tps(xy, synth.tps_residual(uv, synth.BASE_SEED + 2))
    windows = [(0, 777, 0, 1001), (0, 777, 1001, 2000), (777, 2000, 0, 63), (777, 2000, 63, 2000)]
    hip.eval_mode(hip.EVAL_DIRECT)
    try:
        direct = hip.interpolate(g, fit)
        pieces = torch.empty_like(direct)
        for (r0, r1, c0, c1) in windows:
            hip.interpolate(g, fit, window=(r0, r1, c0, c1), out=pieces[r0:r1, c0:c1])
        assert torch.equal(direct, pieces)  # direct sum: ragged windows, same cells, same bits
    finally:
        hip.eval_mode(hip.EVAL_AUTO)
    # default (far-field-interpolated at this size): tiles start at the window origin, so windows agree
    # with the full grid and with the direct sum to rounding rather than bit for bit
    full = hip.interpolate(g, fit)
    for (r0, r1, c0, c1) in windows:
        hip.interpolate(g, fit, window=(r0, r1, c0, c1), out=pieces[r0:r1, c0:c1])
    scale = direct.abs().max().item()
    assert (full - pieces).abs().max().item() < 1e-11 * scale
    assert (full - direct).abs().max().item() < 1e-11 * scale
    del direct, pieces
    m = {"knots": fit.knots, "c": fit.c, "d": fit.d, "center": fit.center, "scale": fit.scale}
    for r0 in (0, 1234, 1999):
        want = cbind.tps_eval_grid(m, g.xmin, g.ymax, g.xres, g.yres, r0, r0 + 1, 0, 2000, threads=8)
        assert np.abs(full[r0:r0 + 1].cpu().numpy() - want).max() < 1e-10 * np.abs(want).max()


@pytest.mark.parametrize("n", [5000, 20000])
def test_fitted_spline_satisfies_its_normal_equations(hip, n):
    """(K + lambda I) c + T d = y and T'c = 0, checked through the GPU evaluation at the knots:
    f(x_i) = y_i - lambda c_i.  n = 20 000 is BASELINE config 5's station count (3.2 GB Gram matrix)."""
    from machisplin_amd import synth
    g = synth.grid(20000, 20000)
    xy, rows, cols, uv = synth.stations(g, n, synth.BASE_SEED + 5)
    y = synth.tps_residual(uv, synth.BASE_SEED + 5)
    lam = 1e-4
    fit = hip.Tps(xy, y, lambda_=lam)
    f = fit.predict(xy)
    assert np.abs(f - (y - lam * fit.c)).max() < 1e-7 * np.abs(y).max()
    T = np.column_stack([np.ones(n), fit.knots])
    assert np.abs(T.T @ fit.c).max() < 1e-7 * np.abs(fit.c).max() * n ** 0.5
    if n == 5000:  # GCV route lands on a lambda whose surface obeys the same identity
        fg = hip.Tps(xy, y)
        assert np.abs(fg.predict(xy) - (y - fg.lambda_ * fg.c)).max() < 1e-7 * np.abs(y).max()
        assert 3.0 < fg.eff_df < n


def test_cfg3_ensemble_linearity_on_the_full_grid(hip):
    """pred = (sum_k w_k pred_k) / wt.tot: the fused Step-2 loop equals the members evaluated one by one."""
    import torch
    from machisplin_amd import synth
    side = 10000
    g, stack, planes = _cfg3_stack(hip, side)
    xy, rows, cols, uv = synth.stations(g, 2000, 11)
    cov = planes[:, torch.from_numpy(rows).cuda(), torch.from_numpy(cols).cuda()].cpu().numpy().astype(np.float64).T
    X = np.column_stack([np.nan_to_num(cov), xy])
    y = synth.response(X, uv, 2)
    params = synth.ensemble_params(X, y, 4, n_gbm_trees=300, n_rf_trees=20, which="gnmv")
    mods = [hip.models.from_param_dict(p) for p in params]
    wts, tot = [0.22, 0.12, 0.18, 0.41], 1.23
    fused = hip.ensemble_predict(stack, mods, wts, tot)
    acc = None
    for m, w in zip(mods, wts):
        pk = hip.predict(stack, m) * w
        acc = pk if acc is None else acc + pk
    ref = (acc.cpu().numpy()) / tot
    got = fused.cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    assert np.allclose(got, ref, rtol=1e-15, atol=0, equal_nan=True)


def test_cfg5_gcv_fit_of_20000_stations(hip):
    """BASELINE configs[4]: the GCV route at n = 20 000 (band reduction of the 3.2 GB projected Gram matrix, host
    search on the band, back-transform).  Properties: the normal equations hold at the chosen lambda, the
    coefficients agree with the Cholesky route run at that same lambda, T'c = 0, and the criterion value the fit
    reports is not above its neighbours' (fixed-lambda fits a factor 2 either side have a larger GCV -- computed
    from their own residuals and the trace identity is not available there, so the check is on the surface: the
    GCV fit's residual sum of squares lies between theirs)."""
    from machisplin_amd import synth
    n = 20000
    g = synth.grid(20000, 20000)
    xy, rows, cols, uv = synth.stations(g, n, synth.BASE_SEED + 5)
    y = synth.tps_residual(uv, synth.BASE_SEED + 5)
    fg = hip.Tps(xy, y)
    lam = fg.lambda_
    assert lam > 0 and 3.0 < fg.eff_df < n and np.isfinite(fg.gcv)
    f = fg.predict(xy)
    assert np.abs(f - (y - lam * fg.c)).max() < 1e-7 * np.abs(y).max()
    T = np.column_stack([np.ones(n), fg.knots])
    assert np.abs(T.T @ fg.c).max() < 1e-7 * np.abs(fg.c).max() * n ** 0.5
    fc = hip.Tps(xy, y, lambda_=lam)                      # MFMA Cholesky at the same lambda
    assert np.abs(fg.c - fc.c).max() < 1e-7 * np.abs(fc.c).max()
    assert np.abs(fg.d - fc.d).max() < 1e-7 * np.abs(fc.d).max()
    rss = lambda fit: float(np.sum((y - fit.predict(xy)) ** 2))
    lo, hi = hip.Tps(xy, y, lambda_=lam / 2), hip.Tps(xy, y, lambda_=lam * 2)
    assert rss(lo) < rss(fg) < rss(hi)                    # RSS is monotone in lambda
    # GCV(lambda) = (RSS / n) / (1 - trA / n)^2 with the reported effective degrees of freedom
    assert abs(rss(fg) / n / (1.0 - fg.eff_df / n) ** 2 - fg.gcv) < 1e-4 * fg.gcv


def test_cfg5_grid_evaluation_20000_squared(hip):
    """20 000 knots over 4e8 cells: the far-field-interpolated evaluation of the whole grid against the direct sum
    on sampled row bands (both HIP) and against the C oracle's direct sum on sampled rows."""
    import torch
    from machisplin_amd import synth
    n, side = 20000, 20000
    g = synth.grid(side, side)
    xy, rows, cols, uv = synth.stations(g, n, synth.BASE_SEED + 5)
    fit = hip.Tps(xy, synth.tps_residual(uv, synth.BASE_SEED + 5), lambda_=1e-4)
    full = hip.interpolate(g, fit)
    assert fit.eval_plan()[0] > 0                          # the far-field path was taken
    # 20 000 terms of either sign cancel by orders of magnitude in a fitted spline (sum |c phi| ~ 1e4 max|f| at
    # lambda = 1e-4), and both GPU sums carry the table log's 2e-14 of that: 1e-8 of max|f| here, two orders inside
    # the north star's 1e-6
    scale = full.abs().max().item()
    hip.eval_mode(hip.EVAL_DIRECT)
    try:
        for r0 in (0, 7777, side - 32):
            band = hip.interpolate(g, fit, window=(r0, r0 + 32, 0, side))
            assert (band - full[r0:r0 + 32]).abs().max().item() < 1e-8 * scale
            del band
    finally:
        hip.eval_mode(hip.EVAL_AUTO)
    m = {"knots": fit.knots, "c": fit.c, "d": fit.d, "center": fit.center, "scale": fit.scale}
    for r0 in (3, 12345):
        want = cbind.tps_eval_grid(m, g.xmin, g.ymax, g.xres, g.yres, r0, r0 + 1, 0, side, threads=16)
        assert np.abs(full[r0:r0 + 1].cpu().numpy() - want).max() < 1e-8 * scale
    del full
    torch.cuda.empty_cache()


def test_cfg5_five_covariate_ensemble_rows(hip):
    """cfg5's Step 2: 5 covariate planes (p = 7) over 20 000 x 20 000 cells with a 20 000-station forest (trees of
    ~12 000 nodes: the BIG form of the level-synchronous walk).  A row band of the fused ensemble against the C
    oracle, and band vs full-grid window consistency of the tree members (same cells, same bits)."""
    import torch
    from machisplin_amd import synth
    side, n = 20000, 20000
    g = synth.grid(side, side)
    planes, nodata = synth.covariates(g, 5, synth.BASE_SEED + 5, dtype="f32", nodata_frac=0.001)
    stack = hip.RasterStack(g, planes, nodata)
    xy, rows, cols, uv = synth.stations(g, n, synth.BASE_SEED + 5)
    cov = planes[:, torch.from_numpy(rows).cuda(), torch.from_numpy(cols).cuda()].cpu().numpy().astype(np.float64).T
    X = np.column_stack([np.nan_to_num(cov), xy])
    y = synth.response(X, uv, 5)
    params = synth.ensemble_params(X, y, 5, n_gbm_trees=600, n_rf_trees=12)
    assert np.diff(params[4]["tree_offsets"]).max() > 8191
    mods = [hip.models.from_param_dict(p) for p in params]
    _, wts, tot = hip.models.select_weights(synth.OPTX_WEIGHTS)
    r0, r1 = 9990, 10022
    band = hip.ensemble_predict(stack, mods, wts, tot, window=(r0, r1, 0, side)).cpu().numpy()
    host = planes[:, r0:r0 + 2].cpu().numpy().astype(np.float64)
    xs, ys = otps.cell_centres(g.xmin, g.ymax, g.xres, g.yres, side, side, r0, r0 + 2)
    Xg = oe.stack_predictors(host, (xs, ys))
    want = cbind.ensemble(params, wts, tot, Xg, threads=16).reshape(2, side)
    assert np.array_equal(np.isnan(band[:2]), np.isnan(want))
    assert np.nanmax(np.abs(band[:2] - want)) < 1e-10 * np.nanmax(np.abs(want))
    # a tree member over a larger window holds the same bits on the shared rows (window origin independence)
    for k in (0, 4):
        a = hip.predict(stack, mods[k], window=(r0, r1, 0, side))
        b = hip.predict(stack, mods[k], window=(r0 - 1000, r1 + 24, 0, side))
        if k == 4:
            assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(b[1000:1000 + (r1 - r0)]))
            # ... and the block-subtree kernel (the default for these trees) the bits of the whole-tree kernel and of the node walk
            for env in ({"MHS_RF_KERNEL": "compact"}, {"MHS_TREES_GENERIC": "1"}):
                os.environ.update(env)
                try:
                    c = hip.predict(stack, mods[k], window=(r0 - 1000, r1 + 24, 0, side))
                finally:
                    for e in env:
                        del os.environ[e]
                assert torch.equal(torch.isnan(b), torch.isnan(c)) and torch.equal(torch.nan_to_num(b), torch.nan_to_num(c)), env
        else:      # gbm: whole 16-row tiles (anchored to the grid; round 4: 16 x 16 cells) hold the same bits, the window's clipped edge tiles the same to rounding
            lo, hi = -r0 % 16, (r1 - r0) - r1 % 16
            assert hi - lo >= 16
            assert torch.equal(torch.nan_to_num(a[lo:hi]), torch.nan_to_num(b[1000 + lo:1000 + hi]))
            assert float((torch.nan_to_num(a) - torch.nan_to_num(b[1000:1000 + (r1 - r0)])).abs().max()) <= 1e-13 * float(a.abs().max())


def test_cfg4_one_unit_full_size(hip):
    """BASELINE configs[3] at full size, ONE (tile, layer) unit: the south-west user tile of machisplin.tiles.create
    (10 000 x 10 000 grid, 2 x 2 tiles, feather.d = 50: 5 025 x 5 025 cells, ~1 260 of the 5 000 stations), smooth
    members (gam, nnet, earth, ksvm), the reference's own 4 x 4 Step-3 tiles inside.  (a) The sharded driver's unit
    (HipTileOps.tile_layer, what TileShardedMltps runs on a rank) equals the hand-written chain mltps_predict bit for bit,
    in the one-call form and in the tile-by-tile form that reports the tiles' lambdas; (b) sampled rows of the ensemble
    plane match the C oracle; (c) inside Step-3 tile (1, 1), clear of every overlap strip, the residual surface is that
    tile's own spline: the oracle's fit of the tile's stations at the GPU's lambda, evaluated on the fit raster's cells."""
    import torch
    from machisplin_amd import sharded, synth
    from oracle import tiles as ot
    side, n, seed = 10000, 5000, synth.BASE_SEED + 4
    g = synth.grid(side, side)
    xy, rows, cols, uv = synth.stations(g, n, seed)
    tiles = hip.tiles.tiles_create(g, xy, out_ncol=2, out_nrow=2, feather_d=50)
    t = 0
    r0, r1, c0, c1 = (int(v) for v in tiles["win"][t])
    assert (r1 - r0, c1 - c0) == (5025, 5025)
    planes, nodata = synth.covariates(g, 3, seed, dtype="f32", window=(r0, r1, c0, c1))
    tg = tiles["geom"][t]
    stack = hip.RasterStack(tg, planes, nodata)
    X = np.column_stack([synth.covariates_at(g, 3, seed, rows, cols), xy])
    resp = synth.response(X, uv, seed) + 3.0 * np.sin(2 * uv[:, 0])
    iv = np.column_stack([xy, resp])
    sel = tiles["dat"][t]
    assert 1000 < len(sel) < 1600
    _, wts, tot = hip.models.select_weights([0.22, 0.12, 0.18, 0.41], labels="gnmv")
    params = synth.ensemble_params(X[sel], resp[sel], seed + t, which="gnmv")
    models = [hip.models.from_param_dict(p) for p in params]
    fitted = {t: {0: {"models": models, "weights": wts, "wt_total": tot}}}
    ops = sharded.HipTileOps(g, tiles, lambda tt: stack, iv, fitted, tile_edge=1500)
    unit = torch.empty((r1 - r0, c1 - c0), dtype=torch.float64, device="cuda")
    rsq_m, rsq_f = ops.tile_layer(t, 0, unit)
    keep = hip.mltps.complete_cases(stack, iv[sel])
    hand = hip.mltps_predict(stack, iv[sel, :2], iv[sel, 2], models, wts, tot, tile_edge=1500, keep=keep)
    assert torch.equal(unit, hand["final"]) and rsq_m == hand["rsq_model"] and rsq_f == hand["rsq_final"]
    assert 0.5 < rsq_m < rsq_f
    byt = hip.mltps_predict(stack, iv[sel, :2], iv[sel, 2], models, wts, tot, tile_edge=1500, keep=keep, tps_info=True)
    assert torch.equal(byt["final_tps"], hand["final_tps"])
    info = byt["tps_info"]
    assert (info["nRx"], info["nCx"]) == (4, 4)
    # (b) ensemble rows vs the C oracle
    rr = 2500
    host = planes[:, rr:rr + 2].cpu().numpy().astype(np.float64)
    xs, ys = otps.cell_centres(tg.xmin, tg.ymax, tg.xres, tg.yres, tg.nrow, tg.ncol, rr, rr + 2)
    want = cbind.ensemble(params, wts, tot, oe.stack_predictors(host, (xs, ys)), 8).reshape(2, -1)
    got = hand["pred_elev"][rr:rr + 2].cpu().numpy()
    assert np.abs(got - want).max() < 1e-11 * np.abs(want).max()
    # (c) the interior of Step-3 tile (row 1, column 1) of the 4 x 4 layout
    og = ot.Geom(tg.xmin, tg.ymax, tg.xres, tg.yres, tg.nrow, tg.ncol)
    nRx, nCx, fw, kw = ot.step3_windows(og, 1500)
    h = 1 * nCx + 1
    kr0, kr1, kc0, kc1 = kw[h]
    # keep windows overlap their neighbours by 2 x 2.5 % of a tile (~63 cells) and Step 4 feathers inside those strips:
    # 200 cells in from every edge the plane is this tile's spline alone
    ir0, ir1, ic0, ic1 = kr0 + 200, kr1 - 200, kc0 + 200, kc1 - 200
    for k in range(nRx * nCx):
        if k != h:
            o = kw[k]
            assert o[1] <= ir0 or o[0] >= ir1 or o[3] <= ic0 or o[2] >= ic1
    assert ir1 - ir0 > 700 and ic1 - ic0 > 700
    knots, res = hand["residuals"][:, 1:], None
    Xs, _, _ = hip.mltps.station_predictors(stack, iv[sel, :2])
    res = hip.mltps.ensemble_residuals(models, wts, tot, Xs[keep], iv[sel, 2][keep])
    ssel = ot.stations_in_window(og, fw[h], knots, None)
    assert len(ssel) == info["tile_n"][h]
    m = otps.fit(knots[ssel], res[ssel], lam=info["lambda"][h])
    gf = ot.window_geom(og, fw[h])
    rows_o = (ir0 + 7, (ir0 + ir1) // 2, ir1 - 9)
    for ro in rows_o:
        ref = cbind.tps_eval_grid(m, gf.xmin, gf.ymax, gf.xres, gf.yres, ro - fw[h][0], ro - fw[h][0] + 1, ic0 - fw[h][2], ic1 - fw[h][2], threads=8)
        got = hand["final_tps"][ro, ic0:ic1].cpu().numpy()
        assert np.abs(got - ref.ravel()).max() < 1e-9 * np.abs(ref).max(), ro


def test_cfg3_reference_tiled_step3_batch_equals_lanes_on_1e8_cells(hip, monkeypatch):
    """BASELINE config 3 with Step 3 the way the reference computes it at this size (7 x 7 tiles of 131-224 stations,
    V73:656-897), full grid: the one-launch batch (round 6) against the tile-by-tile lane route of rounds 1-5 -- every tile's
    spline agrees to 1e-10, hence the mosaicked and feathered surface; and the surface interpolates the smooth residual."""
    import torch
    from machisplin_amd import synth
    g = synth.grid(10000, 10000)
    xy, rows, cols, uv = synth.stations(g, 5000, synth.BASE_SEED + 3)
    resid = synth.tps_residual(uv, synth.BASE_SEED + 3)
    got = hip.tps_residual_surface(g, xy, resid, tile_edge=1500)
    monkeypatch.setenv("MHS_TILES_BATCH", "0")
    want = hip.tps_residual_surface(g, xy, resid, tile_edge=1500)
    monkeypatch.delenv("MHS_TILES_BATCH")
    assert bool(torch.isfinite(got).all())
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 1e-10 * scale
    # at the stations the smoothing spline stays within the noise of the residual (sd 0.1) of its data
    at = got[torch.from_numpy(rows).cuda(), torch.from_numpy(cols).cuda()].cpu().numpy()
    assert np.sqrt(np.mean((at - resid) ** 2)) < 0.15
