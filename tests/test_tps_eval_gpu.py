"""GPU parity: HIP thin-plate-spline evaluation (through the C ABI) vs the oracle's
predict.Krig restatement, same coefficients, same cell centres."""
import numpy as np
import pytest

from conftest import synth_stations
from oracle import tps as otps

pytestmark = pytest.mark.gpu

# |sum_j c_j phi_j| suffers cancellation (sum|terms| / |sum| ~ 1e2..1e3), so the bound is
# stated against the magnitude of the surface: 1e-10 of max|f| (north-star asks 1e-6).
RTOL = 1e-10


def _check(hip, n, nrow, ncol, seed, window=None, lam=None):
    g = hip.Geometry(-78.0, -5.0, 1.0 / 1200, 1.0 / 1200, nrow, ncol)
    xy, y = synth_stations(n, seed, g if n <= nrow * ncol else None)
    m = otps.fit(xy, y, lam=lam)
    t = hip.Tps.from_coef(m["knots"], m["c"], m["d"], m["lambda"], m["center"], m["scale"])
    r0, r1, c0, c1 = window or (0, nrow, 0, ncol)
    got = hip.interpolate(g, t, window=window).cpu().numpy()
    want = otps.predict_grid(m, g.xmin, g.ymax, g.xres, g.yres, nrow, ncol, r0, r1, c0, c1)
    assert got.shape == want.shape
    err = np.abs(got - want).max() / np.abs(want).max()
    assert err < RTOL, err
    return m, t


@pytest.mark.parametrize("n,nrow,ncol", [(12, 5, 7), (200, 48, 64), (813, 130, 257), (64, 1, 1)])
def test_grid_matches_oracle(hip, n, nrow, ncol):
    _check(hip, n, nrow, ncol, seed=n)


def test_window_and_strided_output(hip):
    import torch
    g = hip.Geometry(-78.0, -5.0, 1.0 / 1200, 1.0 / 1200, 300, 400)
    xy, y = synth_stations(150, 3, g)
    m = otps.fit(xy, y)
    t = hip.Tps.from_coef(m["knots"], m["c"], m["d"], m["lambda"], m["center"], m["scale"])
    hip.eval_mode(hip.EVAL_DIRECT)   # the far-field path tiles from the window origin: equal to rounding only
    try:
        full = hip.interpolate(g, t)
        big = torch.full((300, 512), float("nan"), dtype=torch.float64, device=full.device)
        win = (17, 203, 33, 390)
        hip.interpolate(g, t, window=win, out=big[17:203, 33:390])
        torch.cuda.synchronize()
    finally:
        hip.eval_mode(hip.EVAL_AUTO)
    assert torch.equal(big[17:203, 33:390], full[17:203, 33:390])  # same cells, same bits
    assert torch.isnan(big[:, :33]).all() and torch.isnan(big[:17]).all()
    assert torch.isnan(big[:, 390:]).all() and torch.isnan(big[203:]).all()


@pytest.mark.parametrize("n,nrow,ncol,xres,yres,window", [
    (813, 400, 600, 1.0 / 1200, 1.0 / 1200, None),           # square cells, stations spread over the grid
    (500, 700, 300, 1.0 / 600, 1.0 / 1200, None),            # anisotropic cells and a tall grid
    (300, 500, 500, 1.0 / 1200, 1.0 / 1200, (40, 371, 100, 437)),   # window: most knots lie outside it
])
def test_far_field_path_equals_direct_sum(hip, n, nrow, ncol, xres, yres, window):
    """The far-field-interpolated evaluation (csrc/tps_eval.hip) against the direct sum of the same handle
    and against the oracle: equal to rounding, not merely to the 1e-6 of the north star."""
    g = hip.Geometry(-78.0, -5.0, xres, yres, nrow, ncol)
    xy, y = synth_stations(n, n + 1, g)
    m = otps.fit(xy, y)
    t = hip.Tps.from_coef(m["knots"], m["c"], m["d"], m["lambda"], m["center"], m["scale"])
    try:
        hip.eval_mode(hip.EVAL_DIRECT)
        direct = hip.interpolate(g, t, window=window).cpu().numpy()
        hip.eval_mode(hip.EVAL_FAR_FIELD)
        far = hip.interpolate(g, t, window=window).cpu().numpy()
    finally:
        hip.eval_mode(hip.EVAL_AUTO)
    auto = hip.interpolate(g, t, window=window).cpu().numpy()
    r0, r1, c0, c1 = window or (0, nrow, 0, ncol)
    want = otps.predict_grid(m, g.xmin, g.ymax, g.xres, g.yres, nrow, ncol, r0, r1, c0, c1)
    scale = np.abs(want).max()
    assert not np.array_equal(far, direct)                     # really a different summation
    # the bound that holds for any coefficients is relative to S = sum_j |c_j phi_j| (a fitted spline's
    # terms cancel by 1e3..1e4), like the rounding error of the direct sum itself
    rr, cc = np.arange(r0, r1, 7), np.arange(c0, c1, 7)
    x = (g.xmin + (cc + 0.5) * g.xres - m["center"][0]) / m["scale"][0]
    yy = (g.ymax - (rr + 0.5) * g.yres - m["center"][1]) / m["scale"][1]
    S = 0.0
    for j in range(n):
        d2 = (x[None, :] - m["knots"][j, 0]) ** 2 + (yy[:, None] - m["knots"][j, 1]) ** 2
        with np.errstate(divide="ignore", invalid="ignore"):
            S = S + np.abs(m["c"][j]) * np.abs(np.where(d2 > 0, d2 * np.log(d2), 0.0)) * (0.5 / (8 * np.pi))
    # 1e-13 S: both paths carry the table log's 2e-14 absolute error per term (devmath.h), at different
    # points; the interpolation itself contributes ~1e-15 S
    assert np.abs(far - direct).max() <= 1e-13 * S.max(), (np.abs(far - direct).max(), S.max())
    assert np.abs(far - direct).max() / scale < 1e-10
    assert np.abs(far - want).max() / scale < RTOL
    assert np.abs(auto - want).max() / scale < RTOL


def test_eval_mode_and_plan_reporting(hip):
    """mhs_tps_eval_mode / mhs_tps_eval_plan: small windows and few knots take the direct sum whatever the mode,
    the plan reports what the last evaluation did, bad modes are refused."""
    g = hip.Geometry(-78.0, -5.0, 1.0 / 1200, 1.0 / 1200, 512, 512)
    xy, y = synth_stations(400, 21, g)
    m = otps.fit(xy, y, lam=1e-3)
    t = hip.Tps.from_coef(m["knots"], m["c"], m["d"], m["lambda"], m["center"], m["scale"])
    try:
        hip.eval_mode(hip.EVAL_FAR_FIELD)
        hip.interpolate(g, t)
        tc, tr, node_pairs, cell_pairs = t.eval_plan()
        assert tc >= 64 and tc % 64 == 0 and tr >= 16 and tr % 16 == 0
        ntiles = -(-512 // tc) * -(-512 // tr)
        assert 0 < node_pairs <= ntiles * 256 * 400 and 0 < cell_pairs < 512 * 512 * 400
        hip.interpolate(g, t, window=(0, 512, 0, 40))          # narrower than one tile column block: direct
        assert t.eval_plan() == (0, 0, 0, 0)
        hip.eval_mode(hip.EVAL_DIRECT)
        hip.interpolate(g, t)
        assert t.eval_plan() == (0, 0, 0, 0)
        few = hip.Tps.from_coef(m["knots"][:20], m["c"][:20], m["d"], m["lambda"], m["center"], m["scale"])
        hip.eval_mode(hip.EVAL_FAR_FIELD)
        hip.interpolate(g, few)
        assert few.eval_plan() == (0, 0, 0, 0)                 # 20 knots: not worth tiling
        with pytest.raises(hip.MhsError):
            hip.eval_mode(7)
    finally:
        hip.eval_mode(hip.EVAL_AUTO)


def test_far_field_with_every_knot_outside_the_window(hip):
    """A band far from the stations: every knot sits in the rim bins, the whole sum is interpolated."""
    g = hip.Geometry(-78.0, -5.0, 1.0 / 1200, 1.0 / 1200, 1200, 400)
    rng = np.random.default_rng(8)
    rows, cols = rng.integers(0, 150, 300), rng.integers(0, 400, 300)          # stations in the top 150 rows only
    cells = np.unique(rows * 400 + cols)
    rows, cols = np.divmod(cells, 400)
    xy = np.column_stack([g.x_from_col(cols), g.y_from_row(rows)])
    y = np.sin(0.01 * cols) + 0.1 * rng.standard_normal(cells.size)
    m = otps.fit(xy, y, lam=1e-2)
    t = hip.Tps.from_coef(m["knots"], m["c"], m["d"], m["lambda"], m["center"], m["scale"])
    win = (900, 1200, 0, 400)
    try:
        hip.eval_mode(hip.EVAL_FAR_FIELD)
        far = hip.interpolate(g, t, window=win).cpu().numpy()
        assert t.eval_plan()[0] > 0 and t.eval_plan()[3] == 0              # no (cell, near knot) pair at all
    finally:
        hip.eval_mode(hip.EVAL_AUTO)
    want = otps.predict_grid(m, g.xmin, g.ymax, g.xres, g.yres, 1200, 400, *win)
    assert np.abs(far - want).max() / np.abs(want).max() < RTOL


@pytest.mark.parametrize("seed", range(6))
def test_far_field_equals_direct_on_random_geometries(hip, seed):
    """Random cell sizes (anisotropic), grid shapes, windows and station layouts (clustered, partly outside the
    window): the far-field-interpolated sum against the direct sum of the same handle."""
    rng = np.random.default_rng(1000 + seed)
    nrow, ncol = int(rng.integers(200, 900)), int(rng.integers(200, 900))
    xres, yres = 1.0 / rng.integers(300, 2400), 1.0 / rng.integers(300, 2400)
    g = hip.Geometry(-78.0, -5.0, xres, yres, nrow, ncol)
    n = int(rng.integers(60, 700))
    # clustered stations: a mixture of a uniform cloud and two blobs
    u = np.vstack([rng.uniform(0, 1, (n // 2, 2)), rng.normal([0.2, 0.7], 0.05, (n // 4, 2)),
                   rng.normal([0.8, 0.3], 0.1, (n - n // 2 - n // 4, 2))]).clip(0.001, 0.999)
    cells = np.unique((u[:, 1] * nrow).astype(int) * ncol + (u[:, 0] * ncol).astype(int))
    rows, cols = np.divmod(cells, ncol)
    xy = np.column_stack([g.x_from_col(cols), g.y_from_row(rows)])
    y = np.sin(5 * u[:len(cells), 0]) + rng.standard_normal(len(cells)) * 0.2
    t = hip.Tps(xy, y)
    r0 = int(rng.integers(0, nrow // 3)); r1 = int(rng.integers(2 * nrow // 3, nrow + 1))
    c0 = int(rng.integers(0, ncol // 3)); c1 = int(rng.integers(2 * ncol // 3, ncol + 1))
    win = (r0, r1, c0, c1)
    try:
        hip.eval_mode(hip.EVAL_DIRECT)
        direct = hip.interpolate(g, t, window=win).cpu().numpy()
        hip.eval_mode(hip.EVAL_FAR_FIELD)
        far = hip.interpolate(g, t, window=win).cpu().numpy()
        used = t.eval_plan()[0] > 0
    finally:
        hip.eval_mode(hip.EVAL_AUTO)
    assert np.isfinite(far).all()
    assert np.abs(far - direct).max() <= 1e-9 * np.abs(direct).max(), (used, t.eval_plan())


def test_points_match_oracle_and_knot_coincidence(hip):
    g = hip.Geometry(-78.0, -5.0, 1.0 / 1200, 1.0 / 1200, 200, 200)
    xy, y = synth_stations(300, 11, g)
    m = otps.fit(xy, y)
    t = hip.Tps.from_coef(m["knots"], m["c"], m["d"], m["lambda"], m["center"], m["scale"])
    rng = np.random.default_rng(5)
    pts = np.vstack([xy, np.column_stack([rng.uniform(-78.1, -77.7, 1000), rng.uniform(-5.3, -4.9, 1000)])])
    got = t.predict(pts)  # first 300 points sit exactly on knots (r2 == 0)
    want = otps.predict_points(m, pts)
    assert np.abs(got - want).max() / np.abs(want).max() < RTOL


def test_exactly_linear_residual_gives_plane(hip):
    """G4 analytic KAT: c == 0 => the surface is the plane d0 + d1 u + d2 v."""
    g = hip.Geometry(0.0, 10.0, 0.01, 0.01, 100, 120)
    rng = np.random.default_rng(1)
    kn = rng.uniform(0, 1, (50, 2))
    t = hip.Tps.from_coef(kn, np.zeros(50), [1.5, -2.0, 0.25], 0.0, [0.0, 9.0], [1.2, 1.0])
    got = hip.interpolate(g, t).cpu().numpy()
    u = (g.x_from_col(np.arange(120)) - 0.0) / 1.2
    v = (g.y_from_row(np.arange(100)) - 9.0) / 1.0
    want = 1.5 - 2.0 * u[None, :] + 0.25 * v[:, None]
    assert np.abs(got - want).max() < 1e-13


def test_bad_arguments_are_rejected(hip):
    g = hip.Geometry(0.0, 1.0, 0.1, 0.1, 10, 10)
    t = hip.Tps.from_coef(np.random.rand(5, 2), np.zeros(5), [0, 0, 0], 0.0, [0, 0], [1, 1])
    with pytest.raises(hip.MhsError):
        hip.interpolate(g, t, window=(0, 11, 0, 10))
    with pytest.raises(hip.MhsError):
        hip.Tps.from_coef(np.random.rand(5, 2), np.zeros(5), [0, 0, 0], 0.0, [0, 0], [0, 1])
