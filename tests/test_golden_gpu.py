"""GPU vs the COMMITTED fixtures, with no oracle in between: tests/golden/tps_*.npz hold the stations,
the coefficients, both GCV lambdas and two surfaces -- `surf` from the numpy restatement and `surf_scipy` from
scipy.interpolate.RBFInterpolator, the one implementation in this repository's reach that shares no code with
it (tests/golden/make_golden_tps.py).  The HIP fit (fixed lambda = blocked Cholesky; GCV = band reduction + host
search, both modes) and the HIP grid evaluation (both summation paths) are checked against those numbers
directly.  tps_sampling813 carries the reference's own bundled station table (BASELINE.json configs[0])."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = sorted(glob.glob(os.path.join(GOLD, "tps_*.npz")))


def _rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(b).max()


def _geom(hip, z):
    xmin, ymax, res, nrow, ncol = z["geom"]
    return hip.Geometry(float(xmin), float(ymax), float(res), float(res), int(nrow), int(ncol))


def test_fixtures_present():
    assert len(FIXTURES) == 3


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_fixed_lambda_fit_and_grid_match_the_fixture_and_scipy(hip, path):
    z = np.load(path)
    fit = hip.Tps(z["xy"], z["y"], lambda_=float(z["lam"]))
    assert np.array_equal(fit.center, z["center"]) and np.array_equal(fit.scale, z["scale"])
    assert _rel(fit.c, z["c"]) < 1e-8 and _rel(fit.d, z["d"]) < 1e-8
    g = _geom(hip, z)
    scale = np.abs(z["surf_scipy"]).max()
    for mode in (hip.EVAL_DIRECT, hip.EVAL_FAR_FIELD, hip.EVAL_AUTO):
        hip.eval_mode(mode)
        try:
            surf = hip.interpolate(g, fit).cpu().numpy()
        finally:
            hip.eval_mode(hip.EVAL_AUTO)
        assert np.abs(surf - z["surf_scipy"]).max() < 1e-10 * scale, mode   # independent implementation
        assert np.abs(surf - z["surf"]).max() < 1e-10 * scale, mode


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_gcv_fits_land_on_the_fixture_lambdas(hip, path):
    z = np.load(path)
    fg = hip.Tps(z["xy"], z["y"])                          # gcv.Krig: grid + golden section
    lam = float(z["lam_gcv_fields"])
    assert abs(fg.lambda_ - lam) < 1e-8 * lam
    assert abs(fg.gcv - float(z["gcv_fields"])) < 1e-9 * float(z["gcv_fields"])
    assert abs(fg.eff_df - float(z["eff_df_fields"])) < 1e-5 * float(z["eff_df_fields"])
    assert _rel(fg.c, z["c_gcv"]) < 1e-7 and _rel(fg.d, z["d_gcv"]) < 1e-7
    fc = hip.Tps(z["xy"], z["y"], gcv_mode="converged")
    lamc = float(z["lam_gcv_converged"])
    assert abs(fc.lambda_ - lamc) < 1e-6 * lamc
    assert fc.gcv <= fg.gcv * (1 + 1e-12)


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_coefficients_captured_elsewhere_evaluate_to_the_scipy_surface(hip, path):
    """mhs_tps_from_coef: the route a captured fields::Tps object ($c, $d, $knots, $transform) takes."""
    z = np.load(path)
    knots = (z["xy"] - z["center"]) / z["scale"]
    fit = hip.Tps.from_coef(knots, z["c"], z["d"], float(z["lam"]), z["center"], z["scale"])
    surf = hip.interpolate(_geom(hip, z), fit).cpu().numpy()
    assert np.abs(surf - z["surf_scipy"]).max() < 1e-10 * np.abs(z["surf_scipy"]).max()
    pts = z["xy"][:50]
    want = z["y"][:50] - float(z["lam"]) * z["c"][:50]       # (K + lambda I) c + T d = y at the knots
    assert np.abs(fit.predict(pts) - want).max() < 1e-8 * np.abs(z["y"]).max()
