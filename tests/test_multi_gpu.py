"""GPU: several device slots behind the C ABI (csrc/multi.hip; include/machisplin_hip.h "several devices") -- the path a
single-threaded R host takes.  A one-GPU box hosts N slots on GPU 0 (mhs_init_devices with repeated ids): per-slot
contexts, model / spline replicas, one host thread per slot, row bands with slot 0 carrying the fit, the coefficient
hand-over, band evaluation with the whole grid's plan, Step 5 across bands, the stitch (peer copies here: RCCL refuses two
ranks on one device) and the (tile, layer) units with the merge on the layer's owner all run; only the transport differs
from an 8-GPU node.  Every N-slot plane must equal the one-slot plane BIT FOR BIT, and the one-slot plane the existing
single-device Python chain."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NROW, NCOL, N = 333, 420, 700


def _workload(hip, seed=11, nrow=NROW, ncol=NCOL, n=N, dtype="f32"):
    import torch
    from machisplin_amd import synth
    g = synth.grid(nrow, ncol)
    planes, nodata = synth.covariates(g, 3, seed, dtype=dtype)
    xy, rows, cols, uv = synth.stations(g, n, seed)
    cov = planes[:, torch.from_numpy(rows).cuda(), torch.from_numpy(cols).cuda()].cpu().numpy().astype(np.float64).T
    X = np.column_stack([cov, xy])
    resp = synth.response(X, uv, seed)
    params = synth.ensemble_params(X, resp, seed, n_gbm_trees=150, n_rf_trees=8)
    models = [hip.models.from_param_dict(p) for p in params]
    _, weights, wt_total = hip.models.select_weights(synth.OPTX_WEIGHTS)
    return g, planes, nodata, xy, X, resp, models, weights, wt_total


@pytest.fixture(scope="module")
def slots4(hip):
    """Four slots on GPU 0 for this module; one slot again afterwards (other modules only ever use slot 0)."""
    from machisplin_amd import multi
    yield multi.init_devices(4, [0, 0, 0, 0])
    multi.init_devices(1, [0])


@pytest.mark.timeout(900)
@pytest.mark.parametrize("tile_edge", [None, 100])
def test_row_bands_over_slots_equal_the_single_device_chain_bit_for_bit(hip, tile_edge):
    """1, 2 and 4 slots (global fit, and the reference-tiled Step 3 with its tiles dealt over the slots) against
    mltps_predict on one device; with and without a shorter band on slot 0; host-plane call and resident call + gather."""
    import torch
    from machisplin_amd import multi
    multi.init_devices(1, [0])
    g, planes, nodata, xy, X, resp, models, weights, wt_total = _workload(hip)
    stack = hip.RasterStack(g, planes, nodata)
    ref = hip.mltps_predict(stack, xy, resp, models, weights, wt_total, tile_edge=tile_edge)
    torch.cuda.synchronize()
    want = ref["final"].cpu().numpy()
    assert ref["rsq_final"] > ref["rsq_model"]
    host = planes.cpu().numpy()
    for n_slots in (1, 2, 4):
        assert multi.init_devices(n_slots, [0] * n_slots) == [0] * n_slots
        for share in (None, 0.1, 0.0):
            if n_slots == 1 and share is not None:
                continue
            got, info = multi.mltps_grid_multi(g, host, nodata, models, weights, wt_total, X, resp, tile_edge=tile_edge, slot0_share=share)
            assert info["n_slots"] == n_slots and info["used_tps"] and info["collective"] == "none"
            assert sum(b - a for a, b in info["bands"]) == NROW and all(a % 16 == 0 for a, _ in info["bands"][1:])
            # the two R^2 are O(n) host sums: long double accumulators in the library (as R's sum), numpy's pairwise sum in the mirror
            assert info["rsq_model"] == pytest.approx(ref["rsq_model"], rel=1e-13) and info["rsq_final"] == pytest.approx(ref["rsq_final"], rel=1e-13)
            assert np.array_equal(got, want, equal_nan=True), (n_slots, share, np.nanmax(np.abs(got - want)))
        # resident bands, the stitched plane on every slot
        ms = multi.MultiStack(g, host, nodata, slot0_share=0.12 if n_slots > 1 else None)
        info = ms.step(models, weights, wt_total, X, resp, tile_edge=tile_edge, gather=True)
        # one slot: the stitch is ncclAllGather of one rank (binding, communicator and call exercised for real); several slots on one
        # device: RCCL refuses two ranks on a device, the stitch degrades to device copies
        assert info["collective"] == ("rccl-all-gather" if n_slots == 1 else "peer-copies")
        assert np.array_equal(ms.download(), want, equal_nan=True)
        for k in range(n_slots):
            assert np.array_equal(ms.gathered(k), want, equal_nan=True), (n_slots, k)
        ms.free()
    multi.init_devices(1, [0])


@pytest.mark.timeout(600)
def test_block_subtree_forest_on_slot_replicas(hip):
    """A forest whose trees (~3 700 nodes, 6 000 stations) take rf_walk_cbs_kernel: its compact records, subtree table and
    terminal predictions are built per slot on the model's replicas.  Two slots' plane = the one-device chain's, bit for bit."""
    import torch
    from machisplin_amd import multi, synth
    multi.init_devices(1, [0])
    g, planes, nodata, xy, X, resp, models, weights, wt_total = _workload(hip, seed=13, n=6000)
    rf = synth.rf_params(X, resp, 13, n_trees=5)
    assert 3200 < np.diff(rf["tree_offsets"]).max() <= 4095
    models = [hip.models.from_param_dict(rf) if isinstance(m, hip.models.RandomForest) else m for m in models]
    stack = hip.RasterStack(g, planes, nodata)
    ref = hip.mltps_predict(stack, xy, resp, models, weights, wt_total, tile_edge=None)
    torch.cuda.synchronize()
    want = ref["final"].cpu().numpy()
    host = planes.cpu().numpy()
    try:
        multi.init_devices(2, [0, 0])
        got, info = multi.mltps_grid_multi(g, host, nodata, models, weights, wt_total, X, resp, tile_edge=None, slot0_share=0.3)
        assert info["n_slots"] == 2
        assert np.array_equal(got, want, equal_nan=True), np.nanmax(np.abs(got - want))
    finally:
        multi.init_devices(1, [0])


@pytest.mark.timeout(600)
def test_pred_elev_is_returned_when_the_spline_does_not_help(hip, slots4):
    """V73:925-930 across bands: a response the ensemble already explains and pure-noise residuals leave rsq.final <=
    rsq.model, and every slot hands back its pred.elev rows."""
    import torch
    from machisplin_amd import multi
    g, planes, nodata, xy, X, resp, models, weights, wt_total = _workload(hip, seed=5)
    rng = np.random.default_rng(3)
    stack = hip.RasterStack(g, planes, nodata)
    pred = hip.ensemble_predict(stack, models, weights, wt_total).cpu().numpy()
    # a response equal to the ensemble at the stations plus noise the spline cannot use with a huge lambda
    rows, cols = hip.tiles.cells_from_xy(g, xy)
    resp2 = pred[rows, cols] + 1e-3 * rng.standard_normal(rows.size)
    got, info = multi.mltps_grid_multi(g, planes.cpu().numpy(), nodata, models, weights, wt_total, X, resp2, lambda_=1e6)
    ref = hip.mltps_predict(stack, xy, resp2, models, weights, wt_total, tile_edge=None, lambda_=1e6)
    assert info["used_tps"] == (ref["rsq_final"] > ref["rsq_model"])
    assert np.array_equal(got, ref["final"].cpu().numpy(), equal_nan=True)


@pytest.mark.timeout(600)
def test_float64_and_int16_planes_and_nodata_across_slots(hip, slots4):
    import torch
    from machisplin_amd import multi, synth
    for dtype in ("f64", "i16"):
        g = synth.grid(200, 260)
        planes, nodata = synth.covariates(g, 3, 17, dtype=dtype, nodata_frac=0.002 if dtype == "i16" else 0.0)
        xy, rows, cols, uv = synth.stations(g, 400, 17)
        cov = planes[:, torch.from_numpy(rows).cuda(), torch.from_numpy(cols).cuda()].cpu().numpy().astype(np.float64).T
        if dtype == "i16":
            cov[cov == nodata] = np.nan
        ok = ~np.isnan(cov).any(axis=1)
        X = np.column_stack([cov, xy])[ok]
        resp = synth.response(X, uv[ok], 17)
        models = [hip.models.from_param_dict(p) for p in synth.ensemble_params(X, resp, 17, n_gbm_trees=100, n_rf_trees=6)]
        _, weights, wt_total = hip.models.select_weights(synth.OPTX_WEIGHTS)
        stack = hip.RasterStack(g, planes, nodata)
        ref = hip.mltps_predict(stack, xy[ok], resp, models, weights, wt_total, tile_edge=None)
        got, info = multi.mltps_grid_multi(g, planes.cpu().numpy(), nodata, models, weights, wt_total, X, resp)
        assert info["n_slots"] == 4
        assert np.array_equal(got, ref["final"].cpu().numpy(), equal_nan=True), dtype
        if dtype == "i16":
            assert np.isnan(got).sum() > 0


@pytest.mark.timeout(600)
def test_host_plane_pipeline_edge_cases(hip):
    """mhs_mltps_grid_multi's pipeline (sub-bands up under the first members, finished sub-bands down under the last one): ONE
    member (it is first and last at once), a band shorter than a 16-row tile (empty sub-bands), the caller's plane reused, the
    device buffers of another shape discarded and rebuilt, and a failing call that must not leave copies in flight or the
    library unusable."""
    import torch
    from machisplin_amd import multi, _lib
    multi.init_devices(3, [0, 0, 0])
    g, planes, nodata, xy, X, resp, models, weights, wt_total = _workload(hip, seed=23, nrow=97, ncol=130, n=300)
    stack = hip.RasterStack(g, planes, nodata)
    host = planes.cpu().numpy()
    out = np.full((97, 130), -1.0)
    for members in ([5], [0, 5], list(range(len(models)))):          # ksvm alone; gbm + ksvm; all six (synth order b g n m r v)
        mods, wts = [models[k] for k in members], [weights[k] for k in members]
        ref = hip.mltps_predict(stack, xy, resp, mods, wts, wt_total, tile_edge=None)
        got, info = multi.mltps_grid_multi(g, host, nodata, mods, wts, wt_total, X, resp, out=out)
        assert got is out and np.array_equal(out, ref["final"].cpu().numpy(), equal_nan=True), members
        assert sum(b - a for a, b in info["bands"]) == 97 and info["n_slots"] == 3
    # another shape in between: the cached buffers are rebuilt, then rebuilt again
    g2, planes2, nodata2, xy2, X2, resp2, models2, weights2, wt2 = _workload(hip, seed=29, nrow=64, ncol=48, n=120)
    ref2 = hip.mltps_predict(hip.RasterStack(g2, planes2, nodata2), xy2, resp2, models2, weights2, wt2, tile_edge=None)
    got2, _ = multi.mltps_grid_multi(g2, planes2.cpu().numpy(), nodata2, models2, weights2, wt2, X2, resp2)
    assert np.array_equal(got2, ref2["final"].cpu().numpy(), equal_nan=True)
    ref = hip.mltps_predict(stack, xy, resp, models, weights, wt_total, tile_edge=None)
    got, _ = multi.mltps_grid_multi(g, host, nodata, models, weights, wt_total, X, resp)
    assert np.array_equal(got, ref["final"].cpu().numpy(), equal_nan=True)
    # a member with the wrong predictor count fails the call before anything is enqueued ...
    with pytest.raises(_lib.MhsError):
        multi.mltps_grid_multi(g, host, nodata, [hip.models.Gam(np.ones(9))], [1.0], 1.0, X, resp)
    # ... a fit that cannot be done (all stations on one point) fails it on slot 0 while the other slots' copies are in flight
    Xbad = X.copy()
    Xbad[:, -2:] = X[0, -2:]
    with pytest.raises(_lib.MhsError):
        multi.mltps_grid_multi(g, host, nodata, models, weights, wt_total, Xbad, resp)
    got, _ = multi.mltps_grid_multi(g, host, nodata, models, weights, wt_total, X, resp)
    assert np.array_equal(got, ref["final"].cpu().numpy(), equal_nan=True)
    multi.trim()                                     # the kept buffers released: the next call builds them again
    got, _ = multi.mltps_grid_multi(g, host, nodata, models, weights, wt_total, X, resp)
    assert np.array_equal(got, ref["final"].cpu().numpy(), equal_nan=True)
    with pytest.raises(ValueError):
        multi.mltps_grid_multi(g, host, nodata, models, weights, wt_total, X, resp, out=np.zeros((97, 131)))
    multi.init_devices(1, [0])


@pytest.mark.timeout(600)
def test_host_plane_calls_from_several_host_threads(hip):
    """The host-plane calls keep device buffers between calls; several host threads calling at once are served one after the
    other (ctypes releases the GIL inside the call) and every one gets the one-device plane."""
    import threading
    from machisplin_amd import multi
    multi.init_devices(2, [0, 0])
    g, planes, nodata, xy, X, resp, models, weights, wt_total = _workload(hip, seed=31, nrow=160, ncol=200, n=300)
    ref = hip.mltps_predict(hip.RasterStack(g, planes, nodata), xy, resp, models, weights, wt_total, tile_edge=None)["final"].cpu().numpy()
    host = planes.cpu().numpy()
    got, errs = [None] * 4, []
    def call(i):
        try:
            for _ in range(2):
                got[i], _ = multi.mltps_grid_multi(g, host, nodata, models, weights, wt_total, X, resp)
        except Exception as e:          # noqa: BLE001 -- reported below
            errs.append(e)
    ths = [threading.Thread(target=call, args=(i,)) for i in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    for i in range(4):
        assert np.array_equal(got[i], ref, equal_nan=True), i
    multi.init_devices(1, [0])


# ---------------------------------------------------------------- machisplin.tiles.* units (BASELINE config 4) --
T_NROW, T_NCOL, T_N, T_LAYERS = 300, 380, 900, 3


@pytest.mark.timeout(900)
def test_tile_layer_units_over_slots_equal_the_python_chain(hip):
    """mhs_tiles_units_multi (1, 2, 3 and 4 slots on GPU 0) == tiles.create -> mltps_predict per tile and layer ->
    tiles.merge composed in Python on one device, bit for bit, R^2 values included."""
    import torch
    from machisplin_amd import multi, synth
    multi.init_devices(1, [0])
    g = synth.grid(T_NROW, T_NCOL)
    xy, rows, cols, uv = synth.stations(g, T_N, 21)
    tiles = hip.tiles.tiles_create(g, xy, out_ncol=2, out_nrow=2, feather_d=24)
    full, nodata = synth.covariates(g, 3, 21, dtype="f32")
    cov = synth.covariates_at(g, 3, 21, rows, cols)
    Xall = np.column_stack([cov, xy])
    base = synth.response(Xall, uv, 21)
    resp = np.column_stack([base + l + np.sin((2 + l) * uv[:, 0]) for l in range(T_LAYERS)])
    _, wts, tot = hip.models.select_weights([0.22, 0.12, 0.18, 0.41], labels="gnmv")
    units = [[None] * 4 for _ in range(T_LAYERS)]
    want, want_rsq = [], np.zeros((T_LAYERS, 4, 2))
    for l in range(T_LAYERS):
        planes = []
        for t in range(4):
            r0, r1, c0, c1 = (int(v) for v in tiles["win"][t])
            sub = hip.RasterStack(tiles["geom"][t], full[:, r0:r1, c0:c1].contiguous(), nodata)
            sel = tiles["dat"][t]
            Xt, _, _ = hip.mltps.station_predictors(sub, xy[sel])          # the TILE raster's cell-centre LONG / LAT
            ok = ~np.isnan(Xt).any(axis=1)
            models = [hip.models.from_param_dict(p) for p in synth.ensemble_params(Xt[ok], resp[sel, l][ok], 50 + 7 * l + t, which="gnmv")]
            units[l][t] = {"models": models, "weights": wts, "wt_total": tot, "X": Xt[ok], "resp": resp[sel, l][ok]}
            res = hip.mltps_predict(sub, xy[sel], resp[sel, l], models, wts, tot, tile_edge=100)
            planes.append(res["final"].contiguous())
            want_rsq[l, t] = res["rsq_model"], res["rsq_final"]
        want.append(hip.tiles.tiles_merge(g, tiles["win"], planes, in_ncol=2, in_nrow=2).cpu().numpy())
    torch.cuda.synchronize()
    host = full.cpu().numpy()
    for n_slots in (1, 2, 3, 4):
        multi.init_devices(n_slots, [0] * n_slots)
        outs, rsq, info = multi.tiles_units_multi(g, host, nodata, 2, 2, 24, units, T_LAYERS, tile_edge=100)
        assert info["n_slots"] == n_slots and info["n_units"] == 4 * T_LAYERS
        assert np.allclose(rsq, want_rsq, rtol=1e-13, atol=0), n_slots
        for l in range(T_LAYERS):
            assert np.array_equal(outs[l], want[l], equal_nan=True), (n_slots, l)
    # a layer whose merge is skipped stays untouched; smooth members only (tps = False) return pred.elev
    outs, rsq, _ = multi.tiles_units_multi(g, host, nodata, 2, 2, 24, units, T_LAYERS, tps=False, tile_edge=100, merge_layers=[1])
    assert outs[0] is None and outs[2] is None and np.isnan(rsq[:, :, 1]).all()
    # the caller's planes, reused: layers 0 and 2 are written, layer 1's plane is left alone
    mine = [np.full((T_NROW, T_NCOL), -7.0) for _ in range(T_LAYERS)]
    outs, _, _ = multi.tiles_units_multi(g, host, nodata, 2, 2, 24, units, T_LAYERS, tile_edge=100, merge_layers=[0, 2], out=mine)
    assert outs[0] is mine[0] and outs[1] is None and (mine[1] == -7.0).all()
    assert np.array_equal(mine[0], want[0], equal_nan=True) and np.array_equal(mine[2], want[2], equal_nan=True)
    multi.init_devices(1, [0])


@pytest.mark.timeout(300)
def test_slot_bookkeeping_and_errors(hip):
    from machisplin_amd import multi, _lib
    assert multi.init_devices(3, [0, 0, 0]) == [0, 0, 0]
    hip.init(0)                                      # a one-device caller's init leaves the slots alone
    assert multi.device_slots() == [0, 0, 0]
    with pytest.raises(_lib.MhsError):
        multi.init_devices(2, [0, 99])
    assert multi.init_devices(1, [0]) == [0]
    g, planes, nodata, xy, X, resp, models, weights, wt_total = _workload(hip, nrow=64, ncol=80, n=60)
    ms = multi.MultiStack(g, planes.cpu().numpy(), nodata)
    assert ms.bands() == [(0, 64)]
    with pytest.raises(_lib.MhsError):               # a model with the wrong predictor count
        bad = hip.models.Gam(np.ones(9))
        ms.step([bad], [1.0], 1.0, X, resp)
    multi.init_devices(2, [0, 0])
    with pytest.raises(_lib.MhsError):               # the stack was cut for another set of slots
        ms.step(models, weights, wt_total, X, resp)
    multi.init_devices(1, [0])
