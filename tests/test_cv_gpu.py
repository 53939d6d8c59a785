"""GPU parity for SURVEY.md 8f rank 3: hold-out predictions of the fold models (V73:258-319) through
mhs_predict_points against the oracle's predict restatements, both branches of the >4000-row rule."""
import numpy as np
import pytest

from oracle import ensemble as oe

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,nfolds", [(900, 5), (4200, 3)])
def test_cv_residual_columns_match_oracle_and_feed_the_weight_search(hip, n, nfolds):
    from machisplin_amd import cv, synth
    rng = np.random.default_rng(n)
    X = np.column_stack([rng.uniform(76, 4668, n), rng.uniform(-1, 877, n), rng.uniform(-207, 152, n),
                         rng.uniform(-78.0, -77.0, n), rng.uniform(-6.0, -5.0, n)])
    uv = np.column_stack([(X[:, 3] + 78.0), (X[:, 4] + 6.0)])
    y = synth.response(X, uv, 9)
    kfolds = rng.permutation(np.arange(n) % nfolds) + 1
    fold_params, fold_models = [], []
    for v in range(1, nfolds + 1):
        train = np.flatnonzero(kfolds == v) if n > 4000 else np.flatnonzero(kfolds != v)   # V73:228-232
        params = synth.ensemble_params(X[train], y[train], 100 + v, n_gbm_trees=200, n_rf_trees=15)
        fold_params.append(dict(zip(cv.ORDER_ALL, params)))
        fold_models.append({lab: hip.models.from_param_dict(p) for lab, p in zip(cv.ORDER_ALL, params)})
    got = cv.cv_residuals(fold_models, X, y, kfolds)
    want = oe.cv_residuals(fold_params, X, y, kfolds)
    assert got.shape == want.shape == ((nfolds - 1) * n if n > 4000 else n, 6)
    assert np.abs(got - want).max() <= 1e-11 * np.abs(y).max()
    p_gpu = cv.optx_weights(got)
    p_ref = cv.optx_weights(want)
    assert np.allclose(p_gpu[0], p_ref[0], atol=1e-6) and p_gpu[1] == p_ref[1]


@pytest.mark.parametrize("n,nfolds", [(600, 10), (4100, 4)])
def test_linear_member_fitted_per_fold_on_the_device(hip, n, nfolds):
    """V73:252 for the `g` member: the fold models come from mhs_lm_fit; their hold-out residual column equals the
    oracle's (least squares on the fold's training rows, predict on its hold-out rows)."""
    from machisplin_amd import cv
    rng = np.random.default_rng(7 * n)
    X = np.column_stack([rng.uniform(76, 4668, n), rng.uniform(-1, 877, n), rng.uniform(-207, 152, n),
                         rng.uniform(-78.0, -77.0, n), rng.uniform(-6.0, -5.0, n)])
    y = 0.004 * X[:, 0] - 0.01 * X[:, 1] + 3.0 * X[:, 3] + rng.standard_normal(n)
    kfolds = rng.permutation(np.arange(n) % nfolds) + 1
    gams = cv.fit_linear_folds(X, y, kfolds)
    assert len(gams) == nfolds
    got = cv.cv_residuals([{"g": m} for m in gams], X, y, kfolds, labels="g")
    fold_params = []
    for v in range(1, nfolds + 1):
        tr = np.flatnonzero(kfolds == v) if n > 4000 else np.flatnonzero(kfolds != v)
        fold_params.append({"g": oe.lm_model(oe.lm_fit(X[tr], y[tr]))})
    want = oe.cv_residuals(fold_params, X, y, kfolds, labels="g")
    assert got.shape == want.shape and np.abs(got - want).max() <= 1e-8 * np.abs(y).max()
