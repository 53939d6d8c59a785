"""The linear member's fit on the device (mhs_lm_fit: Householder QR of [1 X]) against the oracle's least squares
(oracle/ensemble.py lm_fit, V73:252 / V73:600)."""
import numpy as np
import pytest

from oracle import ensemble as oe

pytestmark = pytest.mark.gpu


def _design(n, p, seed):
    rng = np.random.default_rng(seed)
    X = np.column_stack([rng.normal(1500.0, 400.0, n), rng.uniform(0.0, 35.0, n), rng.normal(8.0, 2.0, n),
                         rng.uniform(-78.0, -76.0, n), rng.uniform(-7.0, -5.0, n), rng.normal(0, 1, n),
                         rng.normal(0, 1, n)])[:, :p]
    beta = rng.normal(0, 1, p + 1)
    y = beta[0] + X @ beta[1:] + 0.3 * rng.standard_normal(n)
    return X, y


@pytest.mark.parametrize("n,p", [(150, 5), (5000, 5), (5000, 7), (20000, 7), (12, 5)])
def test_lm_fit_matches_least_squares(hip, n, p):
    from machisplin_amd.models import Gam
    X, y = _design(n, p, 100 + n + p)
    m = Gam.fit(X, y)
    ref = oe.lm_fit(X, y)
    scale = np.abs(ref).max()
    assert np.abs(m.coefficients - ref).max() < 1e-9 * scale   # LONG/LAT vary by 2 in 77: cond ~ 1e5
    # and the device model built from it predicts the fitted values
    fitted = m.predict_points(X)
    assert np.abs(fitted - (ref[0] + X @ ref[1:])).max() < 1e-9 * np.abs(y).max()
    # normal equations: the residual is orthogonal to [1 X]
    r = y - fitted
    A = np.column_stack([np.ones(n), X])
    assert np.abs(A.T @ r).max() < 1e-7 * np.abs(A.T @ y).max()


def test_lm_fit_rejects_bad_designs(hip):
    from machisplin_amd import _lib
    from machisplin_amd.models import Gam
    X, y = _design(200, 5, 3)
    Xd = X.copy(); Xd[:, 4] = 2.0 * Xd[:, 3]            # collinear
    with pytest.raises(_lib.MhsError) as e:
        Gam.fit(Xd, y)
    assert e.value.code == _lib.ERR_NUMERIC
    Xn = X.copy(); Xn[7, 2] = np.nan
    with pytest.raises(_lib.MhsError) as e:
        Gam.fit(Xn, y)
    assert e.value.code == _lib.ERR_INVALID
    with pytest.raises(_lib.MhsError):
        Gam.fit(X[:5], y[:5])                            # fewer rows than coefficients


def test_lm_fit_rank_test_is_per_column_like_dqrdc2(hip):
    """lm's tolerance compares |R_jj| with the column's own original norm: a covariate in tiny units beside
    elevation in metres is NOT rank deficiency (lm_fit.hip)."""
    from machisplin_amd.models import Gam
    rng = np.random.default_rng(9)
    n = 400
    X = np.column_stack([rng.normal(1500.0, 400.0, n), 1e-5 * rng.standard_normal(n), rng.uniform(-78, -76, n)])
    y = 3.0 + 0.01 * X[:, 0] + 2e4 * X[:, 1] + 0.5 * X[:, 2] + 0.1 * rng.standard_normal(n)
    m = Gam.fit(X, y)
    ref = oe.lm_fit(X, y)
    assert np.abs((m.coefficients - ref) / ref).max() < 1e-7
