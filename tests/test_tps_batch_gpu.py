"""GPU parity of the batched small fits (csrc/tps_batch.hip, mhs_tps_fit_many): every spline of a reference-tiled
Step 3 (V73:690-738) fitted by ONE kernel launch, against the oracle's QR + eigen + GCV restatement and against the
library's one-fit-at-a-time route (mhs_tps_fit) on the same stations."""
import numpy as np
import pytest

from conftest import synth_stations
from oracle import tps as otps

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(b).max()


SIZES = [10, 33, 64, 97, 131, 160, 189, 193, 224, 225, 246, 256]


@pytest.mark.parametrize("mode", ["fields", "converged"])
def test_batch_gcv_fits_match_oracle_and_single_fit_route(hip, mode):
    sets = [synth_stations(n, 900 + n) for n in SIZES]
    fits = hip.tps.fit_many([s[0] for s in sets], [s[1] for s in sets], gcv_mode=mode)
    assert len(fits) == len(SIZES)
    for (xy, y), got, n in zip(sets, fits, SIZES):
        assert got is not None and got.n == n
        want = otps.fit(xy, y, gcv_mode=mode)
        tol = 1e-8 if mode == "fields" else 1e-6
        assert abs(got.lambda_ - want["lambda"]) / want["lambda"] < tol, n
        assert abs(got.eff_df - want["eff_df"]) < 1e-5 * want["eff_df"], n
        assert abs(got.gcv - want["gcv"]) < 1e-9 * want["gcv"], n
        ref = otps.fit(xy, y, lam=got.lambda_)
        assert _rel(got.c, ref["c"]) < 1e-8, n
        assert _rel(got.d, ref["d"]) < 1e-8, n
        assert np.array_equal(got.knots, want["knots"])
        # the library's own one-at-a-time route (one-block tridiagonalisation + host search)
        one = hip.Tps(xy, y, gcv_mode=mode)
        assert abs(got.lambda_ - one.lambda_) / one.lambda_ < (1e-10 if mode == "fields" else 1e-6), n
        if mode == "fields":
            assert _rel(got.c, one.c) < 1e-9 and _rel(got.d, one.d) < 1e-9, n
        # the knot records written by the kernel evaluate like the handle built from host coefficients
        rng = np.random.default_rng(n)
        pts = np.column_stack([rng.uniform(-78, -76, 300), rng.uniform(-7, -5, 300)])
        assert _rel(got.predict(pts), otps.predict_points(ref, pts)) < 1e-9, n


def test_batch_fixed_lambda(hip):
    sets = [synth_stations(n, 400 + n) for n in (12, 100, 200, 250)]
    lam = 3e-3
    fits = hip.tps.fit_many([s[0] for s in sets], [s[1] for s in sets], lambda_=lam)
    for (xy, y), got in zip(sets, fits):
        want = otps.fit(xy, y, lam=lam)
        assert got.lambda_ == lam
        assert _rel(got.c, want["c"]) < 1e-8 and _rel(got.d, want["d"]) < 1e-8
        assert abs(got.eff_df - want["eff_df"]) < 1e-8 * want["eff_df"]


def test_batch_replicates_mixed_sizes_and_failures(hip):
    """Replicated locations collapse inside the batch; a fit too large for a workgroup (n > 256) and one too small
    (n < 8) take mhs_tps_fit's route in the same call; a collinear set fails alone (None), the others are returned."""
    xy, y = synth_stations(150, 7)
    xy2 = np.vstack([xy, xy[:9], xy[3:5]])
    y2 = np.concatenate([y, y[:9] + 0.3, y[3:5] - 0.1])
    big = synth_stations(400, 11)
    tiny = synth_stations(6, 12)
    x = np.linspace(0, 1, 30)
    bad = (np.column_stack([x, 2 * x + 1]), np.sin(x))
    fits = hip.tps.fit_many([xy2, big[0], tiny[0], bad[0]], [y2, big[1], tiny[1], bad[1]])
    assert fits[3] is None
    assert fits[0].n == 150 and fits[1].n == 400 and fits[2].n == 6
    for (a, b), got in zip([(xy2, y2), big, tiny], fits[:3]):
        want = otps.fit(a, b)
        assert abs(got.lambda_ - want["lambda"]) / want["lambda"] < 1e-8
        ref = otps.fit(a, b, lam=got.lambda_)
        assert _rel(got.c, ref["c"]) < 1e-8 and _rel(got.d, ref["d"]) < 1e-8


def test_batch_many_more_fits_than_compute_units(hip):
    """600 fits on 256 compute units: workgroups loop over their fits; every result equals the fit done alone."""
    sets = [synth_stations(40 + (k % 7), 5000 + k) for k in range(600)]
    fits = hip.tps.fit_many([s[0] for s in sets], [s[1] for s in sets])
    for k in (0, 1, 255, 256, 257, 511, 599):
        alone = hip.tps.fit_many([sets[k][0]], [sets[k][1]])[0]
        assert fits[k].lambda_ == alone.lambda_ and np.array_equal(fits[k].c, alone.c) and np.array_equal(fits[k].d, alone.d)
    want = otps.fit(*sets[300])
    assert abs(fits[300].lambda_ - want["lambda"]) / want["lambda"] < 1e-8


def test_tiled_surface_batched_route_equals_lane_route(hip, monkeypatch):
    """mhs_tps_surface's tiles through the batch (one fit launch + one pair of evaluation launches) against the same
    tiles fitted and evaluated one by one on the lanes (MHS_TILES_BATCH=0, the route of rounds 1-5): the spline of every
    tile agrees to 1e-10, hence the mosaicked, feathered surface."""
    from machisplin_amd import synth
    g = synth.grid(700, 900)
    xy, rows, cols, uv = synth.stations(g, 1800, 5)
    resid = synth.tps_residual(uv, 5)
    cov1 = np.ones(1800)
    cov1[::41] = np.nan
    got = hip.tps_residual_surface(g, xy, resid, cov1_at_stations=cov1, tile_edge=300).cpu().numpy()
    monkeypatch.setenv("MHS_TILES_BATCH", "0")
    want = hip.tps_residual_surface(g, xy, resid, cov1_at_stations=cov1, tile_edge=300).cpu().numpy()
    monkeypatch.delenv("MHS_TILES_BATCH")
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() <= 1e-10 * np.abs(want).max()
    # fixed lambda takes the same two routes
    got = hip.tps_residual_surface(g, xy, resid, tile_edge=300, lambda_=2e-3).cpu().numpy()
    monkeypatch.setenv("MHS_TILES_BATCH", "0")
    want = hip.tps_residual_surface(g, xy, resid, tile_edge=300, lambda_=2e-3).cpu().numpy()
    assert np.abs(got - want).max() <= 1e-10 * np.abs(want).max()


def test_tiled_surface_mixes_batch_and_lanes(hip):
    """A tile edge far above the reference's: some tiles hold more than 256 stations (lanes), the sparse corner ones
    fewer (batch); the one-call surface equals the Python composition of the same steps bit for bit."""
    from machisplin_amd import synth
    g = synth.grid(600, 800)
    rng = np.random.default_rng(3)
    # dense in the west, sparse in the east
    cells_w = rng.choice(600 * 400, size=1500, replace=False)
    cells_e = rng.choice(600 * 400, size=150, replace=False)
    rows = np.concatenate([cells_w // 400, cells_e // 400]); cols = np.concatenate([cells_w % 400, 400 + cells_e % 400])
    xy = np.column_stack([g.x_from_col(cols), g.y_from_row(rows)])
    u = (xy - xy.min(0)) / (xy.max(0) - xy.min(0))
    resid = np.sin(6 * u[:, 0]) * np.cos(5 * u[:, 1]) + 0.1 * rng.standard_normal(xy.shape[0])
    info = {}
    want = hip.tps_residual_surface(g, xy, resid, tile_edge=400, info=info).cpu().numpy()
    assert max(info["tile_n"]) > 256 and min(n for n in info["tile_n"] if n >= 10) <= 256
    got = hip.tps_residual_surface(g, xy, resid, tile_edge=400).cpu().numpy()
    assert np.array_equal(got, want)


def test_batch_edge_cases_match_oracle(hip):
    """The edges of the batch kernel's range and of the search: 8 stations (the smallest it takes), 256 (the largest), more
    observations than a workgroup holds collapsing to fewer distinct locations, an exactly linear residual (c = 0: the
    criterion's minimum sits at the upper end of the grid), a constant one, strongly anisotropic coordinates."""
    rng = np.random.default_rng(17)
    sets = {}
    xy, y = synth_stations(8, 801); sets["n8"] = (xy, y)
    xy, y = synth_stations(256, 802); sets["n256"] = (xy, y)
    xy, y = synth_stations(230, 803)
    dup = rng.integers(0, 230, 90)
    sets["collapse_320_to_230"] = (np.vstack([xy, xy[dup]]), np.concatenate([y, y[dup] + 0.05 * rng.standard_normal(90)]))
    xy, _ = synth_stations(120, 804); sets["linear"] = (xy, 2.0 + 0.5 * xy[:, 0] - 0.25 * xy[:, 1])
    xy, _ = synth_stations(60, 805); sets["constant"] = (xy, np.full(60, 3.25))
    xy, y = synth_stations(150, 806); xy = xy.copy(); xy[:, 1] = -6.0 + 1e-3 * (xy[:, 1] + 6.0); sets["anisotropic"] = (xy, y)
    names = list(sets)
    fits = hip.tps.fit_many([sets[k][0] for k in names], [sets[k][1] for k in names])
    for k, got in zip(names, fits):
        a, b = sets[k]
        assert got is not None, k
        want = otps.fit(a, b)
        assert got.n == want["knots"].shape[0], k
        if k in ("linear", "constant"):
            # c = 0 whatever lambda is: the surface is the plane (G4 KAT); lambda sits at the grid's end in both
            assert np.abs(got.c).max() < 1e-8 * max(1.0, np.abs(b).max()), k
            assert np.abs(got.predict(a) - b).max() < 1e-8 * max(1.0, np.abs(b).max()), k
            continue
        assert abs(got.lambda_ - want["lambda"]) / want["lambda"] < 1e-8, (k, got.lambda_, want["lambda"])
        ref = otps.fit(a, b, lam=got.lambda_)
        assert _rel(got.c, ref["c"]) < 1e-7 and _rel(got.d, ref["d"]) < 1e-7, k
        assert abs(got.eff_df - want["eff_df"]) < 1e-5 * want["eff_df"], k


def test_tiled_surface_zero_tiles_and_replicated_stations(hip):
    """A surface whose eastern tiles hold fewer than 10 stations (zero tiles, V73:710-721) and whose western ones hold
    replicated stations: the one-call route (batch) equals the tile-by-tile composition bit for bit, and the oracle's flow."""
    from machisplin_amd import synth
    from oracle import tiles as ot
    g = synth.grid(500, 900)
    rng = np.random.default_rng(23)
    cells = rng.choice(500 * 400, size=700, replace=False)
    rows, cols = cells // 400, cells % 400
    xy = np.column_stack([g.x_from_col(cols), g.y_from_row(rows)])
    xy = np.vstack([xy, xy[:40], [[g.x_from_col(np.array([880]))[0], g.y_from_row(np.array([20]))[0]]] * 3])      # replicates; 3 stations far east
    u = (xy - xy.min(0)) / (xy.max(0) - xy.min(0))
    resid = np.sin(6 * u[:, 0]) * np.cos(5 * u[:, 1]) + 0.1 * rng.standard_normal(xy.shape[0])
    info = {}
    want = hip.tps_residual_surface(g, xy, resid, tile_edge=250, info=info).cpu().numpy()
    assert min(info["tile_n"]) < 10 < max(info["tile_n"])
    got = hip.tps_residual_surface(g, xy, resid, tile_edge=250).cpu().numpy()
    assert np.array_equal(got, want)
    assert np.all(got[:, 700:][np.isfinite(got[:, 700:])] == 0.0) or np.abs(got[:, 750:]).max() < 1e-12      # the zero tiles
