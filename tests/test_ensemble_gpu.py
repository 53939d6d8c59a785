"""GPU parity: the six HIP predictors + weighted ensemble (through the C ABI) vs the oracle's
restatement of the CRAN predict methods on the same covariate planes."""
import os

import numpy as np
import pytest

import modelgen
from oracle import ensemble as oe
from oracle import tps as otps

pytestmark = pytest.mark.gpu

KINDS = "bgnmrv"


def _setup(hip, nrow=70, ncol=93, C=3, dtype="f32", nodata_frac=0.0, n=600, seed=5, gbm_trees=300, rf_trees=20):
    from machisplin_amd import synth
    g = synth.grid(nrow, ncol)
    planes, nodata = synth.covariates(g, C, seed, dtype=dtype, nodata_frac=nodata_frac)
    stack = hip.RasterStack(g, planes, nodata)
    host = planes.cpu().numpy().astype(np.float64)
    if not np.isnan(nodata):
        host[host == nodata] = np.nan
    x, y = otps.cell_centres(g.xmin, g.ymax, g.xres, g.yres, nrow, ncol)
    X = oe.stack_predictors(host, (x, y))
    rng = np.random.default_rng(seed)
    ok = np.flatnonzero(~np.isnan(X).any(axis=1))
    idx = rng.choice(ok, n, replace=False)
    Xs = X[idx]
    uv = np.column_stack([(idx % ncol + 0.5) / ncol, (idx // ncol + 0.5) / nrow])
    ys = synth.response(Xs, uv, seed)
    params = synth.ensemble_params(Xs, ys, seed, n_gbm_trees=gbm_trees, n_rf_trees=rf_trees)
    return g, stack, X, Xs, ys, params


def _tol(ref):
    return 1e-11 * np.nanmax(np.abs(ref))


@pytest.mark.parametrize("dtype", ["f32", "f64", "i16"])
def test_each_member_matches_oracle(hip, dtype):
    g, stack, X, Xs, ys, params = _setup(hip, dtype=dtype, nodata_frac=0.01)
    for prm in params:
        m = hip.models.from_param_dict(prm)
        got = hip.predict(stack, m).cpu().numpy().ravel()
        want = oe.predict(prm, X)
        assert np.array_equal(np.isnan(got), np.isnan(want)), prm["kind"]
        assert np.nanmax(np.abs(got - want)) <= _tol(want), (prm["kind"], np.nanmax(np.abs(got - want)))


@pytest.mark.parametrize("dtype", ["f64", "f32", "i16"])
def test_tree_fast_paths_equal_the_generic_walk_bit_for_bit(hip, dtype, monkeypatch):
    """float64 planes -- what terra holds in RAM and the R shim hands over (V73:468-606, MHS_F64) -- take the
    predicate-LUT / level-synchronous kernels with the rank search done in double against double thresholds;
    float32 / int16 planes search in float.  Either way the trees are summed in the same order as the node walk
    (MHS_TREES_GENERIC=1 forces it), so the planes must be identical, NA cells included."""
    import torch
    g, stack, X, Xs, ys, params = _setup(hip, nrow=150, ncol=211, dtype=dtype, nodata_frac=0.01, n=1500, gbm_trees=700,
                                         rf_trees=30)
    for prm in (params[0], params[4]):
        assert prm["kind"] in ("gbm", "rf")
        m = hip.models.from_param_dict(prm)
        monkeypatch.setenv("MHS_GBM_NO_COHERENT", "1")          # gbm's default on grids sums in another order: its own test
        fast = hip.predict(stack, m)
        monkeypatch.delenv("MHS_GBM_NO_COHERENT")
        monkeypatch.setenv("MHS_TREES_GENERIC", "1")
        slow = hip.predict(stack, m)
        monkeypatch.delenv("MHS_TREES_GENERIC")
        assert torch.equal(torch.isnan(fast), torch.isnan(slow)), prm["kind"]
        assert torch.equal(torch.nan_to_num(fast), torch.nan_to_num(slow)), prm["kind"]
        # the same model on another plane type right after: the geometry tables are rebuilt for the other key type
        other = hip.RasterStack(g, stack.planes.to(torch.float64 if dtype != "f64" else torch.float32),
                                float("nan") if dtype != "i16" else -32768.0)
        again = hip.predict(other, m).cpu().numpy().ravel()
        host = other.planes.cpu().numpy().astype(np.float64)
        if dtype == "i16":
            host[host == -32768.0] = np.nan
        x, y = otps.cell_centres(g.xmin, g.ymax, g.xres, g.yres, g.nrow, g.ncol)
        want = oe.predict(prm, oe.stack_predictors(host, (x, y)))
        assert np.array_equal(np.isnan(again), np.isnan(want))
        assert np.nanmax(np.abs(again - want)) <= _tol(want), prm["kind"]


def test_gbm_routes_na_through_missing_nodes(hip):
    g, stack, X, Xs, ys, params = _setup(hip, dtype="f32", nodata_frac=0.05)
    prm = params[0]
    got = hip.predict(stack, hip.models.from_param_dict(prm)).cpu().numpy().ravel()
    assert np.isnan(X).any() and np.isfinite(got).all()
    assert np.abs(got - oe.predict(prm, X)).max() <= _tol(got)


@pytest.mark.parametrize("frac", [0.002, 0.3])
def test_gbm_na_cells_through_the_compacted_list_equal_the_strided_pass(hip, frac, monkeypatch):
    """gbm's MissingNode routing for cells with an NA covariate (gbm_pred; terra::predict(rast_stack, gbm), V73:497): round 4
    walks them from a compacted list instead of in every block that holds one (0.09 % NoData on the reference's rasters cost
    half the kernel).  0.2 % NA: the list; 30 % NA: more than the list holds -> the overflow word hands the window back to
    the strided pass.  Either way the NA cells' values are the node walk's, bit for bit, and the oracle's."""
    import torch
    g, stack, X, Xs, ys, params = _setup(hip, nrow=160, ncol=300, dtype="f32", nodata_frac=frac, n=900, gbm_trees=400, rf_trees=2)
    prm = params[0]
    assert prm["kind"] == "gbm"
    m = hip.models.from_param_dict(prm)
    a = hip.predict(stack, m)
    acc = hip.predict(stack, m, weight=0.5, accumulate=True, out=a.clone())
    monkeypatch.setenv("MHS_TREES_GENERIC", "1")          # the node walk: NA cells through MissingNode in every block that holds one
    b = hip.predict(stack, m)
    monkeypatch.delenv("MHS_TREES_GENERIC")
    na = torch.from_numpy(np.isnan(X).any(axis=1).reshape(g.nrow, g.ncol)).to(a.device)
    assert not torch.isnan(a).any() and na.any() and torch.equal(a[na], b[na])          # (the other cells: the coherent kernel's order)
    assert float((a - b).abs().max()) <= 1e-13 * float(b.abs().max())
    assert torch.allclose(acc, 1.5 * a, rtol=1e-15, atol=0)
    want = oe.predict(prm, X)
    assert np.nanmax(np.abs(a.cpu().numpy().ravel() - want)) <= _tol(want)
    win = hip.predict(stack, m, window=(5, 150, 3, 290))      # another tiling of the coherent kernel: equal to rounding
    assert float((win - a[5:150, 3:290]).abs().max()) <= 1e-13 * float(a.abs().max())


def test_sklearn_fitted_structures(hip):
    """Real fitted trees / SVR / MLP (exported by tests/modelgen.py) rather than synthetic ones."""
    from sklearn.ensemble import GradientBoostingRegressor, RandomForestRegressor
    from sklearn.svm import SVR
    g, stack, X, Xs, ys, _ = _setup(hip, n=400)
    # sklearn's trees compare float32(x) <= threshold: train off the cell centres so that no cell's
    # LONG/LAT coincides (in float32) with a split value, where `<` (gbm) and `<=` would differ
    Xs = Xs.copy()
    Xs[:, 3:] += np.random.default_rng(0).uniform(-3e-4, 3e-4, (Xs.shape[0], 2))
    gbr = GradientBoostingRegressor(n_estimators=80, max_leaf_nodes=6, learning_rate=0.01, subsample=0.5, random_state=1).fit(Xs, ys)
    rf = RandomForestRegressor(n_estimators=15, min_samples_split=6, max_features=1, random_state=2).fit(Xs, ys)
    mu, sd, ym, ysd = Xs.mean(0), Xs.std(0, ddof=1), ys.mean(), ys.std(ddof=1)
    svr = SVR(kernel="rbf", C=1.0, epsilon=0.1, gamma=0.2).fit((Xs - mu) / sd, (ys - ym) / ysd)
    for prm, skl in [(modelgen.gbm_from_sklearn(gbr, 5), gbr.predict(X)), (modelgen.rf_from_sklearn(rf, 5), rf.predict(X)),
                     (modelgen.svr_from_sklearn(svr, mu, sd, ym, ysd), svr.predict((X - mu) / sd) * ysd + ym)]:
        got = hip.predict(stack, hip.models.from_param_dict(prm)).cpu().numpy().ravel()
        assert np.abs(got - oe.predict(prm, X)).max() <= _tol(got), prm["kind"]
        # independent evaluator; deep RF trees hold thousands of LONG/LAT thresholds, a few of which
        # fall between a cell centre and its float32 rounding (what sklearn compares): allow 5 %
        bad = np.abs(got - skl) > 1e-9 * np.abs(skl).max()
        assert bad.mean() <= (0.05 if prm["kind"] == "rf" else 0.0), (prm["kind"], bad.mean())


def test_weighted_ensemble_window_and_accumulate(hip):
    import torch
    g, stack, X, Xs, ys, params = _setup(hip, nodata_frac=0.01)
    kept, wts, tot = hip.models.select_weights([0.31, 0.22, 0.004, 0.18, 0.27, 0.41])
    assert kept == "bgmrv"
    sel = [params[KINDS.index(k)] for k in kept]
    mods = [hip.models.from_param_dict(p) for p in sel]
    want = oe.ensemble(sel, wts, tot, X).reshape(g.nrow, g.ncol)
    got = hip.ensemble_predict(stack, mods, wts, tot).cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.nanmax(np.abs(got - want)) <= _tol(want)
    # a window written into a larger strided buffer touches nothing else
    win = (11, 57, 20, 81)
    big = torch.full((g.nrow, 128), -7.0, dtype=torch.float64, device="cuda")
    hip.ensemble_predict(stack, mods, wts, tot, window=win, out=big[11:57, 20:81])
    b = big.cpu().numpy()
    # (to rounding: gbm sums a cell's trees in an order that depends on which cells share its wave, i.e. on the window)
    assert np.array_equal(np.isnan(b[11:57, 20:81]), np.isnan(got[11:57, 20:81]))
    assert np.nanmax(np.abs(b[11:57, 20:81] - got[11:57, 20:81])) <= 1e-13 * np.nanmax(np.abs(got))
    b[11:57, 20:81] = -7.0
    assert (b == -7.0).all()
    # step-by-step accumulate == the fused loop (V73:471/475 order)
    acc = hip.predict(stack, mods[0], weight=wts[0])
    for m, w in zip(mods[1:], wts[1:]):
        hip.predict(stack, m, weight=w, accumulate=True, out=acc)
    # (torch may divide by a scalar as a reciprocal multiply, so compare to 1 ulp-ish)
    assert np.allclose(acc.cpu().numpy() / tot, got, rtol=1e-15, atol=0, equal_nan=True)


def test_predict_points_gives_station_residual_inputs(hip):
    g, stack, X, Xs, ys, params = _setup(hip)
    mods = []
    for prm in params:
        m = hip.models.from_param_dict(prm)
        mods.append(m)
        got = m.predict_points(Xs)
        want = oe.predict(prm, Xs)
        assert np.abs(got - want).max() <= _tol(want), prm["kind"]
    # res.FINAL in one call (mhs_residual_points) == the member-by-member accumulation of V73:477-620
    wts = [0.31, 0.22, 0.12, 0.18, 0.27, 0.41]
    got = hip.mltps.ensemble_residuals(mods, wts, 1.51, Xs, ys)
    want = None
    for m, w in zip(mods, wts):
        rk = (ys - m.predict_points(Xs)) * w
        want = rk if want is None else want + rk
    assert np.array_equal(got, want / 1.51)
    ref = None
    for prm, w in zip(params, wts):
        rk = (ys - oe.predict(prm, Xs)) * w
        ref = rk if ref is None else ref + rk
    assert np.abs(got - ref / 1.51).max() <= _tol(ys)


def test_seven_predictors_and_host_entry_point(hip):
    import ctypes as C
    from machisplin_amd import _lib
    g, stack, X, Xs, ys, params = _setup(hip, nrow=40, ncol=50, C=5, n=300, gbm_trees=50, rf_trees=5)
    mods = [hip.models.from_param_dict(p) for p in params]
    wts = [0.3, 0.2, 0.1, 0.2, 0.3, 0.4]
    want = oe.ensemble(params, wts, 1.5, X).reshape(40, 50)
    got = hip.ensemble_predict(stack, mods, wts, 1.5).cpu().numpy()
    assert np.nanmax(np.abs(got - want)) <= _tol(want)
    # host-pointer ABI: covariates and output in host memory, a row window
    host = np.ascontiguousarray(stack.planes.cpu().numpy())
    out = np.empty((25, 50))
    hs = (C.c_void_p * 6)(*[m._h for m in mods])
    ws = (C.c_double * 6)(*wts)
    st = _lib.Stack(host.ctypes.data, 5, _lib.F32, 40 * 50, 50, float("nan"))
    gs = g.c_struct()
    _lib.check(_lib.lib().mhs_ensemble_predict(hs, ws, 6, 1.5, C.byref(gs), C.byref(st), 10, 35, 0, 50, out.ctypes.data))
    assert np.array_equal(out, got[10:35], equal_nan=True)


@pytest.mark.parametrize("which,dtype", [("bgnmrv", "f64"), ("gnmv", "f32"), ("rv", "i16"), ("b", "f32")])
def test_host_pointer_ensemble_pipeline_equals_the_resident_call_bitwise(hip, which, dtype, monkeypatch):
    """mhs_ensemble_predict (what the R shim calls, V73:468-606) on a window of >= 16 M cells: the first member launch is
    banded behind the arriving covariate rows, the last one ahead of the departing result rows, the members between run
    once over the window (host_window_pipeline); a single member, or MHS_HOST_BANDS, takes the all-members band pipeline.
    Whatever the route, the plane is the resident one-call plane bit for bit (NA cells included)."""
    import ctypes as C
    import torch
    from machisplin_amd import _lib, synth
    nrow, ncol = 4100, 4000
    g = synth.grid(nrow, ncol)
    planes, nodata = synth.covariates(g, 3, 5, dtype=dtype, nodata_frac=0.001)
    stack = hip.RasterStack(g, planes, nodata)
    xy, rows, cols, uv = synth.stations(g, 500, 5)
    cov = planes[:, torch.from_numpy(rows).cuda(), torch.from_numpy(cols).cuda()].cpu().numpy().astype(np.float64).T
    X = np.column_stack([cov, xy])
    ok = ~(cov == nodata).any(axis=1) if not np.isnan(nodata) else ~np.isnan(cov).any(axis=1)
    X, uv = X[ok], uv[ok]
    params = synth.ensemble_params(X, synth.response(X, uv, 5), 5, n_gbm_trees=64, n_rf_trees=4, which=which)
    models = [hip.models.from_param_dict(p) for p in params]
    wts = [0.1 + 0.07 * k for k in range(len(models))]
    want = hip.ensemble_predict(stack, models, wts, 1.3).cpu().numpy()
    host = np.ascontiguousarray(planes.cpu().numpy())
    out = np.empty((nrow, ncol))
    hs = (C.c_void_p * len(models))(*[m._h for m in models])
    ws = (C.c_double * len(models))(*wts)
    st = _lib.Stack(host.ctypes.data, 3, {"f64": _lib.F64, "f32": _lib.F32, "i16": _lib.I16}[dtype], nrow * ncol, ncol, float(nodata))
    gs = g.c_struct()
    for bands in (None, "3"):
        if bands:
            monkeypatch.setenv("MHS_HOST_BANDS", bands)
        out[:] = -1.0
        _lib.check(_lib.lib().mhs_ensemble_predict(hs, ws, len(models), 1.3, C.byref(gs), C.byref(st), 0, nrow, 0, ncol, out.ctypes.data))
        if bands:
            monkeypatch.delenv("MHS_HOST_BANDS")
        assert np.array_equal(np.isnan(out), np.isnan(want)), bands
        assert np.array_equal(np.nan_to_num(out), np.nan_to_num(want)), bands
    # a window that does not start at the grid's first row or column
    win = (37, 4090, 11, 3990)
    sub = np.empty((win[1] - win[0], win[3] - win[2]))
    _lib.check(_lib.lib().mhs_ensemble_predict(hs, ws, len(models), 1.3, C.byref(gs), C.byref(st), *win, sub.ctypes.data))
    res = hip.ensemble_predict(stack, models, wts, 1.3, window=win).cpu().numpy()
    assert np.array_equal(np.nan_to_num(sub), np.nan_to_num(res))
    # against the full plane: the window's clipped edge tiles give gbm another order of its sum -- to rounding
    cut = want[win[0]:win[1], win[2]:win[3]]
    assert np.array_equal(np.isnan(sub), np.isnan(cut)) and np.nanmax(np.abs(sub - cut)) <= 1e-13 * np.nanmax(np.abs(cut))


@pytest.mark.parametrize("C,trees,n", [(7, 120, 600), (3, 8000, 7000)])
def test_gbm_rank_lut_variants(hip, C, trees, n):
    """The gbm predicate-LUT path works on RANKS of the keys among the model's sorted distinct split values.
    9 predictors exercise the LDS-key kernel (more than the 8 the register-resident kernel holds); 8 000
    trees over 7 000 stations give > 4 096 distinct thresholds for a predictor, so the coarse + fine rank
    search runs."""
    from machisplin_amd import synth
    g, stack, X, Xs, ys, params = _setup(hip, nrow=96, ncol=100, C=C, n=n, gbm_trees=2, rf_trees=1, nodata_frac=0.01)
    prm = synth.gbm_params(Xs, ys, 11, n_trees=trees)
    distinct = max(len(np.unique(prm["split_val"][prm["split_var"] == v].astype(np.float32))) for v in range(C))
    assert distinct > 4096 or trees < 8000
    got = hip.predict(stack, hip.models.from_param_dict(prm)).cpu().numpy().ravel()
    want = oe.predict(prm, X)
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() <= _tol(want)   # one wrong leaf would be ~1e-3 * sd(y)


@pytest.mark.parametrize("dtype,n_splits,window", [("f32", 5, None), ("f64", 5, (3, 37, 40, 1040)), ("i16", 3, None),
                                                   ("f32", 1, (0, 40, 0, 1000))])
def test_gbm_row_tile_kernel_equals_the_other_paths_bit_for_bit(hip, dtype, n_splits, window, monkeypatch):
    """Rows of >= ~240 cells take gbm_lutreg_rt_kernel (round 3): a wave = 256 consecutive cells of one row, the
    tree's splits on LAT and its padding levels evaluated on the scalar unit, the leaf LUT permuted per tree.  Same
    leaf values in the same tree order as the node walk: identical planes, NA cells,
    ragged last tile, windows, trees with fewer than five splits (n_splits < 5: up to two padding levels go to the
    scalar unit, the rest read as never-true vector levels) included."""
    import torch
    from machisplin_amd import synth
    g, stack, X, Xs, ys, params = _setup(hip, nrow=40, ncol=1100, dtype=dtype, nodata_frac=0.01, n=1500, gbm_trees=2, rf_trees=1)
    prm = synth.gbm_params(Xs, ys, 3, n_trees=900, n_splits=n_splits)
    assert (prm["split_var"] == prm["p"] - 1).any()          # some splits on LAT
    m = hip.models.from_param_dict(prm)
    win = window or (0, g.nrow, 0, 1000)
    monkeypatch.setenv("MHS_GBM_NO_COHERENT", "1")          # the default for such rows, gbm_coherent_kernel: next test
    fast = hip.predict(stack, m, window=win)
    monkeypatch.setenv("MHS_TREES_GENERIC", "1")
    slow = hip.predict(stack, m, window=win)
    monkeypatch.delenv("MHS_TREES_GENERIC")
    assert torch.equal(torch.isnan(fast), torch.isnan(slow))
    assert torch.equal(torch.nan_to_num(fast), torch.nan_to_num(slow))
    assert not torch.isnan(fast).any()          # gbm routes NA covariates through its MissingNode children
    # the lane-per-cell register kernel on 100-column pieces of the same window (too narrow for the row tiles)
    r0, r1, c0, c1 = win
    for cc in range(c0, c1, 100):
        piece = hip.predict(stack, m, window=(r0, r1, cc, min(cc + 100, c1)))
        ref = fast[:, cc - c0:min(cc + 100, c1) - c0]
        assert torch.equal(torch.nan_to_num(piece), torch.nan_to_num(ref)), cc
    want = oe.predict(prm, X).reshape(g.nrow, g.ncol)[r0:r1, c0:c1]
    got = fast.cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.nanmax(np.abs(got - want)) <= _tol(want)


@pytest.mark.parametrize("dtype,n_splits,window,smooth", [("f32", 5, None, False), ("f64", 5, (3, 37, 40, 1040), False), ("i16", 3, None, False),
                                                          ("f32", 1, (0, 40, 0, 1000), False), ("f32", 5, None, True), ("f64", 4, (1, 39, 7, 1100), True)])
def test_gbm_coherent_kernel_matches_the_tree_order_kernels_and_the_oracle(hip, dtype, n_splits, window, smooth, monkeypatch):
    """The default for long rows (round 3), gbm_coherent_kernel: per wave of 256 consecutive cells a tree whose splits all fall
    the same way for every cell is summed once per wave, a tree with one split inside the wave's range costs a compare and
    a select per cell, the others the full evaluation.  Same leaves, another order of the sum: equal to the tree-order
    kernels (MHS_GBM_NO_COHERENT=1) to rounding, NA cells (walked through MissingNode) and ragged tiles included.  The
    standard test rasters vary fast (most trees take the full evaluation); `smooth` stretches them so that most trees
    are wave-uniform or single -- all three branches carry weight in one case or the other."""
    import torch
    from machisplin_amd import synth
    if smooth:
        g = synth.grid(40 * 50, 1100 * 9)                     # the test grid is a corner of a grid 50 x 9 times larger
        planes, nodata = synth.covariates(g, 3, 5, dtype=dtype, window=(0, 40, 0, 1100))
        gs = synth.grid(40, 1100)                             # same origin and cell size: the large grid's NW corner
        stack = hip.RasterStack(gs, planes, nodata)
        host = planes.cpu().numpy().astype(np.float64)
        x, y = otps.cell_centres(gs.xmin, gs.ymax, gs.xres, gs.yres, 40, 1100)
        X = oe.stack_predictors(host, (x, y))
        rng = np.random.default_rng(5)
        idx = rng.choice(X.shape[0], 1500, replace=False)
        Xs = X[idx]
        ys = synth.response(Xs, np.column_stack([(idx % 1100 + 0.5) / 1100, (idx // 1100 + 0.5) / 40]), 5)
        g = gs
    else:
        g, stack, X, Xs, ys, params = _setup(hip, nrow=40, ncol=1100, dtype=dtype, nodata_frac=0.01, n=1500, gbm_trees=2, rf_trees=1)
    prm = synth.gbm_params(Xs, ys, 3, n_trees=900, n_splits=n_splits)
    m = hip.models.from_param_dict(prm)
    win = window or (0, g.nrow, 0, 1000)
    r0, r1, c0, c1 = win
    coh = hip.predict(stack, m, window=win)
    monkeypatch.setenv("MHS_GBM_NO_COHERENT", "1")
    ordered = hip.predict(stack, m, window=win)
    monkeypatch.delenv("MHS_GBM_NO_COHERENT")
    assert not torch.isnan(coh).any()                         # gbm routes NA covariates through its MissingNode children
    assert torch.equal(torch.isnan(coh), torch.isnan(ordered))
    scale = float(ordered.abs().max())
    assert float((coh - ordered).abs().max()) <= 1e-13 * scale
    assert smooth or not torch.equal(coh, ordered)            # i.e. the switch did select another kernel
    want = oe.predict(prm, X).reshape(g.nrow, g.ncol)[r0:r1, c0:c1]
    got = coh.cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.nanmax(np.abs(got - want)) <= _tol(want)


def test_gbm_probe_picks_the_coherent_kernel_on_smooth_rasters_and_the_tree_order_kernel_on_noise(hip, monkeypatch):
    """Windows of 2^20 cells and more: a probe (the coherent kernel's classification on 64 tiles and 256 trees) prices the
    coherent kernel against the tree-order row-tile kernel on the device, both are launched and the loser returns at once.
    Smooth rasters (the BASELINE generator) must come out with the coherent kernel's bits (MHS_GBM_FORCE_COHERENT=1),
    white noise with the tree-order kernel's (MHS_GBM_NO_COHERENT=1) -- whose plane is the node walk's."""
    import torch
    from machisplin_amd import synth
    g = synth.grid(4000, 4096)        # a 64-column tile spans 1.6 % of the rasters' extent (cfg3: 0.6 %)
    planes, nodata = synth.covariates(g, 3, 5, dtype="f32")
    gen = torch.Generator(device="cuda"); gen.manual_seed(3)
    noise = planes.clone()
    for k in range(3):
        lo, hi = float(planes[k].min()), float(planes[k].max())
        noise[k] = lo + (hi - lo) * torch.rand(planes[k].shape, device="cuda", generator=gen)
    xy, rows, cols, uv = synth.stations(g, 1500, 5)
    cov = planes[:, torch.from_numpy(rows).cuda(), torch.from_numpy(cols).cuda()].cpu().numpy().astype(np.float64).T
    X = np.column_stack([cov, xy])
    prm = synth.gbm_params(X, synth.response(X, uv, 5), 3, n_trees=300)
    m = hip.models.from_param_dict(prm)
    for pl, picked, other in ((planes, "MHS_GBM_FORCE_COHERENT", "MHS_GBM_NO_COHERENT"), (noise, "MHS_GBM_NO_COHERENT", "MHS_GBM_FORCE_COHERENT")):
        stack = hip.RasterStack(g, pl, nodata)
        auto = hip.predict(stack, m)
        monkeypatch.setenv(picked, "1")
        a = hip.predict(stack, m)
        monkeypatch.delenv(picked)
        monkeypatch.setenv(other, "1")
        b = hip.predict(stack, m)
        monkeypatch.delenv(other)
        assert torch.equal(auto, a), (picked, float((auto - a).abs().max()), float((auto - b).abs().max()))
        assert not torch.equal(auto, b) and float((auto - b).abs().max()) <= 1e-13 * float(b.abs().max())


@pytest.mark.parametrize("n,dtype,ncol,trees", [(1400, "f32", 257, 7), (4600, "f64", 257, 7), (4600, "i16", 257, 7), (1400, "f32", 1100, 7),
                                                (700, "f32", 257, 130), (300, "f32", 257, 520)])
def test_forest_walk_kernels_equal_each_other_and_the_node_walk(hip, n, dtype, ncol, trees, monkeypatch):
    """The forest's three walk kernels -- the loader-wave kernel (rf_walk_ld_kernel, the default where three trees and the keys
    fit: one loader wave stages the trees, even ones through its registers and odd ones by LDS-DMA, fifteen waves of 16 x 16
    cells walk), the double-buffered kernel (MHS_RF_KERNEL=db) and the split-node kernel (MHS_RF_KERNEL=compact) -- each with
    and without the wave-uniform prefix and the early exit (MHS_RF_PLAIN=1: every tree from the root to its full depth), and the
    generic node walk: bit-identical planes.  1 400 stations give trees of ~900 nodes, 4 600 stations ~2 800 (the 24 KB buffer
    stride).  123 rows: a ragged last strip.  Seven trees: a count that is a multiple neither of the two nor of the three
    buffers; 130: the entry registers are reloaded every 64 trees; 520: more than they hold.  (Round 5 removed the kernels
    and switches that lost in rounds 3 and 4 -- single buffer, triple buffer without a loader, two loaders, the compiler's
    level loop, four-walk forms, 64 x 4 strips: their measurements are in profiles/r03_tree_variants.txt and
    profiles/r04_forest_variants.txt, their code in the history at commit 55aadea.)"""
    import torch
    from machisplin_amd import synth
    g, stack, X, Xs, ys, params = _setup(hip, nrow=123, ncol=ncol, dtype=dtype, nodata_frac=0.01, n=n, gbm_trees=2, rf_trees=1)
    prm = synth.rf_params(Xs, ys, 9, n_trees=trees)     # 130: the entry registers are reloaded every 64 trees; 520: more than they hold
    nodes = np.diff(prm["tree_offsets"]).max()
    assert (nodes <= 2048) == (n < 2000) and nodes <= 3072
    m = hip.models.from_param_dict(prm)
    fast = hip.predict(stack, m)
    for envs in ({"MHS_RF_PLAIN": "1"}, {"MHS_RF_KERNEL": "sub"}, {"MHS_RF_KERNEL": "sub", "MHS_RF_PLAIN": "1"}, {"MHS_RF_KERNEL": "db"}, {"MHS_RF_KERNEL": "db", "MHS_RF_PLAIN": "1"},
                 {"MHS_RF_KERNEL": "compact"}, {"MHS_RF_KERNEL": "compact", "MHS_RF_PLAIN": "1"},
                 {"MHS_RF_KERNEL": "cbs"}, {"MHS_RF_KERNEL": "cbs", "MHS_RF_PLAIN": "1"}, {"MHS_RF_KERNEL": "cbs", "MHS_RF_CBS_ROUGH": "0"},
                 {"MHS_RF_KERNEL": "cbs", "MHS_RF_CBS_ROUGH": "1000000"}, {"MHS_TREES_GENERIC": "1"}):
        for e, v in envs.items():
            monkeypatch.setenv(e, v)
        other = hip.predict(stack, m)
        for e in envs:
            monkeypatch.delenv(e)
        assert torch.equal(torch.isnan(fast), torch.isnan(other)), envs
        assert torch.equal(torch.nan_to_num(fast), torch.nan_to_num(other)), envs
    want = oe.predict(prm, X)
    got = fast.cpu().numpy().ravel()
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.nanmax(np.abs(got - want)) <= _tol(want)


def test_forest_with_trees_between_the_loader_wave_and_the_compact_forms(hip, monkeypatch):
    """Trees of 3 201 .. 4 095 nodes (6 000 stations): three of them do not fit LDS (no loader-wave kernel), and the block-subtree
    kernel takes them by default (round 6; the double-buffered kernel before).  Default, double-buffered, whole-tree compact and
    the node walk: the same bits; the oracle to its tolerance."""
    import torch
    from machisplin_amd import synth
    g, stack, X, Xs, ys, params = _setup(hip, nrow=123, ncol=257, dtype="f32", nodata_frac=0.01, n=6000, gbm_trees=2, rf_trees=1)
    prm = synth.rf_params(Xs, ys, 9, n_trees=7)
    nodes = np.diff(prm["tree_offsets"]).max()
    assert 3200 < nodes <= 4095
    m = hip.models.from_param_dict(prm)
    fast = hip.predict(stack, m)
    for envs in ({"MHS_RF_KERNEL": "db"}, {"MHS_RF_KERNEL": "cbs"}, {"MHS_RF_KERNEL": "compact"}, {"MHS_RF_PLAIN": "1"}, {"MHS_TREES_GENERIC": "1"}):
        for e, v in envs.items():
            monkeypatch.setenv(e, v)
        other = hip.predict(stack, m)
        for e in envs:
            monkeypatch.delenv(e)
        assert torch.equal(torch.isnan(fast), torch.isnan(other)), envs
        assert torch.equal(torch.nan_to_num(fast), torch.nan_to_num(other)), envs
    want = oe.predict(prm, X)
    got = fast.cpu().numpy().ravel()
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.nanmax(np.abs(got - want)) <= _tol(want)


@pytest.mark.parametrize("case", range(10))
def test_forest_block_subtree_kernel_on_random_shapes(hip, case, monkeypatch):
    """rf_walk_cbs_kernel pinned (MHS_RF_KERNEL=cbs) on seeded random shapes -- 1 .. 10 covariates (p = 3 .. 12), 40 .. 5 000
    stations (trees of a few dozen to ~3 000 nodes), 1 .. 70 trees, every plane type, 0 .. 5 % NoData, grids whose sides are no
    multiples of the 16 x 16 wave tiles, a window inside the grid, a threshold that sends every / no block through the
    whole-tree loop -- against the node walk, bit for bit, NA cells included."""
    import torch
    from machisplin_amd import synth
    rng = np.random.default_rng(1000 + case)
    C = int(rng.integers(1, 11))
    n = int(rng.choice([40, 150, 700, 2500, 5000]))
    nrow, ncol = int(rng.integers(17, 140)), int(rng.integers(20, 400))
    n = min(n, nrow * ncol // 3)
    dtype = ["f32", "f64", "i16"][case % 3]
    g, stack, X, Xs, ys, params = _setup(hip, nrow=nrow, ncol=ncol, C=C, dtype=dtype, nodata_frac=float(rng.choice([0.0, 0.01, 0.05])), n=n,
                                         seed=20 + case, gbm_trees=2, rf_trees=1)
    prm = synth.rf_params(Xs, ys, 30 + case, n_trees=int(rng.integers(1, 71)))
    m = hip.models.from_param_dict(prm)
    r0, c0 = int(rng.integers(0, nrow - 16)), int(rng.integers(0, ncol - 1))
    win = (r0, int(rng.integers(r0 + 16, nrow + 1)), c0, int(rng.integers(c0 + 1, ncol + 1)))
    monkeypatch.setenv("MHS_TREES_GENERIC", "1")
    want_full, want_win = hip.predict(stack, m), hip.predict(stack, m, window=win)
    monkeypatch.delenv("MHS_TREES_GENERIC")
    monkeypatch.setenv("MHS_RF_KERNEL", "cbs")
    for rough in (None, "0", "1000000"):
        if rough is not None:
            monkeypatch.setenv("MHS_RF_CBS_ROUGH", rough)
        for want, w in ((want_full, None), (want_win, win)):
            got = hip.predict(stack, m, window=w)
            assert torch.equal(torch.isnan(got), torch.isnan(want)), (case, rough, w)
            assert torch.equal(torch.nan_to_num(got), torch.nan_to_num(want)), (case, rough, w)


@pytest.mark.parametrize("C,n,force", [(11, 900, None), (13, 900, None), (11, 5200, "compact"), (16, 700, None)])
def test_forest_with_more_than_twelve_predictors(hip, C, n, force, monkeypatch):
    """p = C + 2 >= 13 predictors (11+ covariate layers, V73:127-138 adds LONG and LAT): the wave-uniform prefix keeps the
    wave's [min, max] ranks for 12 predictors only, so these forests must start their walks at the root (round-3 advisor
    finding: a split on predictor >= 12 used predictor 0's range).  Grid of 123 rows (strips + prefix would be on), every
    walk form that fits, against the oracle and against the generic node walk bit for bit."""
    import torch
    from machisplin_amd import synth
    g, stack, X, Xs, ys, params = _setup(hip, nrow=123, ncol=257, C=C, dtype="f32", nodata_frac=0.01, n=n, gbm_trees=2, rf_trees=1)
    prm = synth.rf_params(Xs, ys, 11, n_trees=9)
    bv = np.asarray(prm["best_var"])[np.asarray(prm["status"]) == -3] - 1        # 0-based split predictors
    assert np.isin(np.arange(12, C + 2), bv).all(), "no split on a predictor >= 12"
    m = hip.models.from_param_dict(prm)
    if force:
        monkeypatch.setenv("MHS_RF_KERNEL", force)
    fast = hip.predict(stack, m)
    if force:
        monkeypatch.delenv("MHS_RF_KERNEL")
    monkeypatch.setenv("MHS_TREES_GENERIC", "1")
    generic = hip.predict(stack, m)
    monkeypatch.delenv("MHS_TREES_GENERIC")
    assert torch.equal(torch.isnan(fast), torch.isnan(generic))
    assert torch.equal(torch.nan_to_num(fast), torch.nan_to_num(generic))
    want = oe.predict(prm, X)
    got = fast.cpu().numpy().ravel()
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.nanmax(np.abs(got - want)) <= _tol(want)


def test_forest_with_trees_larger_than_the_16_bit_lds_addresses(hip):
    """Trees of ~18 000 nodes (30 000 stations): too many for any LDS form (16-bit byte addresses, 16-bit terminal codes), so
    the forest takes the generic node walk (round 5 removed the BIG forms: the split-node kernel covers every BASELINE
    configuration, 12 000-node trees at cfg5 included).  Same results."""
    from machisplin_amd import synth
    g, stack, X, Xs, ys, params = _setup(hip, nrow=256, ncol=256, C=3, n=30000, gbm_trees=2, rf_trees=1)
    prm = synth.rf_params(Xs, ys, 5, n_trees=3)
    assert np.diff(prm["tree_offsets"]).max() > 8191
    got = hip.predict(stack, hip.models.from_param_dict(prm)).cpu().numpy().ravel()
    want = oe.predict(prm, X)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.nanmax(np.abs(got - want)) <= _tol(want)


@pytest.mark.parametrize("dtype", ["f32", "f64", "i16"])
def test_forest_in_the_compact_form_equals_the_node_walk(hip, dtype):
    """Trees of ~7 000 nodes (12 000 stations): beyond the double-buffered kernel's 4 095, within the COMPACT form
    (split nodes only in LDS, terminals as codes): by default the block-subtree kernel (rf_walk_cbs_kernel: per block of 80 x 48
    cells only the subtrees its cells can reach are staged, several trees at a time), pinned the whole-tree kernel
    (MHS_RF_KERNEL=compact).  Same leaves in the same order as the node walk and the oracle."""
    from machisplin_amd import synth
    g, stack, X, Xs, ys, params = _setup(hip, nrow=200, ncol=300, C=5, n=12000, gbm_trees=2, rf_trees=1, nodata_frac=0.01, dtype=dtype)
    prm = synth.rf_params(Xs, ys, 6, n_trees=5)
    nodes = np.diff(prm["tree_offsets"]).max()
    assert 4095 < nodes and 8 * (nodes // 2) + nodes <= 65535
    model = hip.models.from_param_dict(prm)
    got = hip.predict(stack, model).cpu().numpy().ravel()
    want = oe.predict(prm, X)
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.isnan(got).any()
    assert np.nanmax(np.abs(got - want)) <= _tol(want)
    # MHS_RF_CBS_ROUGH: the LDS slots per tree above which a block takes whole trees -- 0: every block does, 1000000: none
    for envs in ({"MHS_RF_PLAIN": "1"}, {"MHS_TREES_GENERIC": "1"}, {"MHS_RF_KERNEL": "compact"}, {"MHS_RF_KERNEL": "compact", "MHS_RF_PLAIN": "1"},
                 {"MHS_RF_CBS_ROUGH": "0"}, {"MHS_RF_CBS_ROUGH": "1000000"}, {"MHS_RF_CBS_ROUGH": "1000000", "MHS_RF_PLAIN": "1"}):
        os.environ.update(envs)
        try:
            other = hip.predict(stack, model).cpu().numpy().ravel()
        finally:
            for e in envs:
                del os.environ[e]
        assert np.array_equal(got, other, equal_nan=True), envs
    # a window that starts inside a wave tile and is narrower than a block of tiles: same cells
    win = hip.predict(stack, model, window=(37, 150, 21, 94)).cpu().numpy()
    assert np.array_equal(win, got.reshape(200, 300)[37:150, 21:94], equal_nan=True)


@pytest.mark.parametrize("C,dtype", [(1, "f32"), (3, "f64"), (5, "f32"), (9, "i16")])
def test_ksvm_window_and_point_paths_match_the_oracle(hip, C, dtype):
    """ksvm with p = 3, 5, 7, 11 predictors, NA cells, positive and negative alphas: the whole grid, a window of it
    (a cell's value must not depend on the launch geometry) and the stations' point path, against the oracle."""
    g, stack, X, Xs, ys, params = _setup(hip, nrow=61, ncol=83, C=C, n=333, dtype=dtype, nodata_frac=0.02, gbm_trees=2, rf_trees=1)
    prm = params[KINDS.index("v")]
    assert (np.asarray(prm["alpha"]) > 0).any() and (np.asarray(prm["alpha"]) < 0).any()
    model = hip.models.from_param_dict(prm)
    want = oe.predict(prm, X)
    got = hip.predict(stack, model).cpu().numpy().ravel()
    win = hip.predict(stack, model, window=(7, 40, 11, 70)).cpu().numpy()
    pts = model.predict_points(Xs[:100])
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.isnan(got).any()
    assert np.nanmax(np.abs(got - want)) <= _tol(want)
    assert np.array_equal(win, got.reshape(61, 83)[7:40, 11:70], equal_nan=True)
    assert np.abs(pts - oe.predict(prm, Xs[:100])).max() <= _tol(want)


@pytest.mark.parametrize("dtype,C", [("f32", 3), ("f64", 3), ("i16", 3), ("f32", 5)])
def test_ksvm_row_tile_kernel_alone_matches_the_oracle(hip, dtype, C, monkeypatch):
    """svr_rt_kernel on its own (the round-3 verdict's item 4): 560 columns = two full 192-cell waves and a ragged third
    (176 cells), 1 % NoData cells inside the waves, against the oracle at 1e-11 of the prediction; a window whose first
    column is not a multiple of 192; and the lane-per-cell kernel (MHS_SVR_NO_ROWTILE=1) at 1e-12 -- the row-tile kernel
    folds the wave's largest q = sigma |x|^2 / 700 into the per-wave term and scales back with one exponential per cell, so
    the two agree to rounding, not bitwise."""
    g, stack, X, Xs, ys, params = _setup(hip, nrow=37, ncol=560, C=C, n=333, dtype=dtype, nodata_frac=0.01, gbm_trees=2, rf_trees=1)
    prm = params[KINDS.index("v")]
    model = hip.models.from_param_dict(prm)
    want = oe.predict(prm, X)
    got = hip.predict(stack, model).cpu().numpy().ravel()
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.isnan(got).any()
    assert np.nanmax(np.abs(got - want)) <= _tol(want)
    win = hip.predict(stack, model, window=(3, 30, 7, 552)).cpu().numpy()       # 545 columns from column 7: tiles shifted by 7
    ref = want.reshape(37, 560)[3:30, 7:552]
    assert np.array_equal(np.isnan(win), np.isnan(ref))
    assert np.nanmax(np.abs(win - ref)) <= _tol(want)
    monkeypatch.setenv("MHS_SVR_NO_ROWTILE", "1")
    plain = hip.predict(stack, model).cpu().numpy().ravel()
    monkeypatch.delenv("MHS_SVR_NO_ROWTILE")
    assert np.array_equal(np.isnan(got), np.isnan(plain))
    assert not np.array_equal(got, plain, equal_nan=True), "the row-tile kernel did not run (identical planes)"
    assert np.nanmax(np.abs(got - plain)) <= 1e-12 * np.nanmax(np.abs(plain))


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_ksvm_row_tile_kernel_keeps_the_addition_where_a_wave_spreads(hip, dtype, monkeypatch):
    """A covariate JUMP inside a wave: with q = sigma |x~|^2 / 700 spreading more than 0.5 over the wave's 192 cells the
    row-tile kernel must keep the per-pair addition (folding would flush terms the cell's back-scaling cannot restore).
    Plane 0 carries a spike of 24 standard deviations on three columns (250..252, inside the second wave) of rows 0..19 --
    too few cells to move the model's centre and scale -- and sigma = 2: the spread is ~2 * 24^2 / 700 = 1.6, and a wave that
    folded anyway would flush the terms of its ordinary cells.  Rows 20.. have no spike (those waves fold).  Against the oracle
    at 1e-11."""
    import torch
    from machisplin_amd import synth
    nrow, ncol, C = 41, 560, 3
    g = synth.grid(nrow, ncol)
    planes, nodata = synth.covariates(g, C, 5, dtype=dtype, nodata_frac=0.01)
    sd0 = float(torch.nan_to_num(planes[0].double()).std())
    planes[0, :20, 250:253] += 24.0 * sd0
    stack = hip.RasterStack(g, planes, nodata)
    host = planes.cpu().numpy().astype(np.float64)
    x, y = otps.cell_centres(g.xmin, g.ymax, g.xres, g.yres, nrow, ncol)
    X = oe.stack_predictors(host, (x, y))
    rng = np.random.default_rng(5)
    ok = np.flatnonzero(~np.isnan(X).any(axis=1))
    idx = rng.choice(ok, 400, replace=False)
    Xs = X[idx]
    uv = np.column_stack([(idx % ncol + 0.5) / ncol, (idx // ncol + 0.5) / nrow])
    prm = synth.svr_params(Xs, synth.response(Xs, uv, 5), 5)
    prm["sigma"] = 2.0
    xt = (X - prm["x_center"]) / prm["x_scale"]
    q = 2.0 / 700.0 * (xt * xt).sum(1).reshape(nrow, ncol)
    assert np.nanmax(q[3, 192:384]) - np.nanmin(q[3, 192:384]) > 0.5 and np.nanmax(q[30, 192:384]) - np.nanmin(q[30, 192:384]) < 0.5
    model = hip.models.from_param_dict(prm)
    want = oe.predict(prm, X)
    got = hip.predict(stack, model).cpu().numpy().ravel()
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.nanstd(want) > 1e-3 * np.nanmax(np.abs(want))           # the kernel sums are not all flushed to a constant
    assert np.nanmax(np.abs(got - want)) <= _tol(want)
    monkeypatch.setenv("MHS_SVR_NO_ROWTILE", "1")
    plain = hip.predict(stack, model).cpu().numpy().ravel()
    monkeypatch.delenv("MHS_SVR_NO_ROWTILE")
    assert np.nanmax(np.abs(got - plain)) <= 1e-12 * np.nanmax(np.abs(plain))
    # the unfolded waves add exactly what the lane-per-cell kernel adds: bit-identical there
    a, b = got.reshape(nrow, ncol), plain.reshape(nrow, ncol)
    assert np.array_equal(a[:20, 192:384], b[:20, 192:384], equal_nan=True)


def test_loaders_reject_malformed_models(hip):
    with pytest.raises(hip.MhsError):
        hip.models.Gbm(0.0, [0, 2], [0, -1], [1.0, 2.0], [5, 0], [1, 0], [1, 0], p=5)  # child out of range
    with pytest.raises(hip.MhsError):
        hip.models.Earth([1.0, 2.0], [[0, 0], [3, 0]], [[0, 0], [0, 0]])  # bad dir code
    with pytest.raises(hip.MhsError):
        hip.models.RandomForest([0, 1], [0], [0], [-3], [9], [0.0], [1.0], p=5)  # bestvar out of range
    g, stack, *_ = _setup(hip, nrow=8, ncol=8, n=40, gbm_trees=2, rf_trees=1)
    with pytest.raises(ValueError):
        hip.predict(stack, hip.models.Gam([1.0, 2.0, 3.0]))  # p != layers + 2


@pytest.mark.parametrize("which", ["bgnmrv", "gnmv", "nm", "gm", "bnr"])
def test_fused_small_members_equal_member_by_member_accumulation(hip, which):
    """gam, nnet and earth are consecutive in the reference's model order; the library evaluates a run of them in
    one pass over the planes (mhs_members_predict_dev).  Same arithmetic, same accumulation order: the planes are
    identical to member-by-member mhs_predict_dev calls, NA cells included."""
    import torch
    g, stack, X, Xs, ys, params = _setup(hip, nodata_frac=0.01, gbm_trees=40, rf_trees=4)
    sel = [params[KINDS.index(k)] for k in which]
    mods = [hip.models.from_param_dict(p) for p in sel]
    wts = [0.31, 0.22, 0.12, 0.18, 0.27, 0.41][:len(mods)]
    fused = hip.models.members_predict(stack, mods, wts)
    acc = hip.predict(stack, mods[0], weight=wts[0])
    for m, w in zip(mods[1:], wts[1:]):
        hip.predict(stack, m, weight=w, accumulate=True, out=acc)
    assert torch.equal(torch.isnan(fused), torch.isnan(acc))
    assert torch.equal(torch.nan_to_num(fused), torch.nan_to_num(acc)), which
    # accumulate = True adds to what the plane holds
    again = hip.models.members_predict(stack, mods[1:], wts[1:], accumulate=True, out=hip.predict(stack, mods[0], weight=wts[0]))
    assert torch.equal(torch.nan_to_num(again), torch.nan_to_num(acc))
