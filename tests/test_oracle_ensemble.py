"""CPU: pin the ensemble oracle against scikit-learn evaluators of the same structures and
against hand-computed known answers.  The CRAN packages themselves cannot
run here: parity unpinned vs R."""
import numpy as np
import pytest

import modelgen
from oracle import ensemble as oe


def _data(n=400, seed=0):
    rng = np.random.default_rng(seed)
    X = np.column_stack([rng.uniform(76, 4668, n), rng.uniform(-1, 877, n), rng.uniform(-207, 152, n),
                         rng.uniform(-78, -76, n), rng.uniform(-7, -5, n)])
    y = 250 - 0.0055 * X[:, 0] + 3 * np.sin(X[:, 3]) + 0.01 * X[:, 2] + 0.3 * rng.standard_normal(n)
    return X, y


def test_gbm_oracle_vs_sklearn():
    from sklearn.ensemble import GradientBoostingRegressor
    X, y = _data()
    gbr = GradientBoostingRegressor(n_estimators=60, max_leaf_nodes=6, learning_rate=0.05, subsample=0.5, random_state=1).fit(X, y)
    m = modelgen.gbm_from_sklearn(gbr, 5)
    Xt, _ = _data(300, 9)
    assert np.allclose(oe.predict(m, Xt), gbr.predict(Xt), rtol=0, atol=1e-10)
    # NA routing: a NaN covariate takes the MissingNode (the split node's own mean), never NaN
    Xn = Xt.copy()
    Xn[::3, 0] = np.nan
    out = oe.predict(m, Xn)
    assert np.isfinite(out).all()
    assert np.allclose(out[1::3], gbr.predict(Xt)[1::3], atol=1e-10)


def test_rf_oracle_vs_sklearn():
    from sklearn.ensemble import RandomForestRegressor
    X, y = _data()
    rf = RandomForestRegressor(n_estimators=25, min_samples_split=6, max_features=1, random_state=2).fit(X, y)
    m = modelgen.rf_from_sklearn(rf, 5)
    Xt, _ = _data(300, 8)
    assert np.allclose(oe.predict(m, Xt), rf.predict(Xt), rtol=0, atol=1e-10)
    Xn = Xt.copy()
    Xn[5, 2] = np.nan
    assert np.isnan(oe.predict(m, Xn)[5]) and np.isfinite(oe.predict(m, Xn)[6])


def test_svr_oracle_vs_sklearn():
    from sklearn.svm import SVR
    X, y = _data()
    mu, sd = X.mean(0), X.std(0, ddof=1)
    ym, ys = y.mean(), y.std(ddof=1)
    svr = SVR(kernel="rbf", C=1.0, epsilon=0.1, gamma=0.21).fit((X - mu) / sd, (y - ym) / ys)
    m = modelgen.svr_from_sklearn(svr, mu, sd, ym, ys)
    Xt, _ = _data(200, 7)
    ref = svr.predict((Xt - mu) / sd) * ys + ym
    assert np.allclose(oe.predict(m, Xt), ref, rtol=0, atol=1e-10)


def test_nnet_oracle_vs_sklearn_and_saturation():
    from sklearn.neural_network import MLPRegressor
    X, y = _data()
    Xs = (X - X.mean(0)) / X.std(0)
    ymin = y.min()
    ysc = (y - ymin).max()
    mlp = MLPRegressor(hidden_layer_sizes=(10,), activation="logistic", max_iter=300, random_state=3).fit(Xs, (y - ymin) / ysc)
    m = modelgen.nnet_from_sklearn(mlp, ysc, ymin)
    assert np.allclose(oe.predict(m, Xs), mlp.predict(Xs) * ysc + ymin, rtol=0, atol=1e-9)
    # nnet.c clamps the logistic exactly outside [-15, 15]
    assert oe._nnet_sigmoid(np.array([-15.1]))[0] == 0.0 and oe._nnet_sigmoid(np.array([15.1]))[0] == 1.0
    assert 0 < oe._nnet_sigmoid(np.array([-14.9]))[0] < 1e-6


def test_hand_built_models():
    # 3-node gbm stump with NA routing
    g = oe.gbm_model(10.0, [0, 4], [1, -1, -1, -1], [5.0, -1.0, 2.0, 0.5], [1, 0, 0, 0], [2, 0, 0, 0], [3, 0, 0, 0])
    X = np.array([[0.0, 4.9], [0.0, 5.0], [0.0, np.nan]])
    assert np.allclose(oe.predict(g, X), [9.0, 12.0, 10.5])  # x < 5 left; 5 is NOT < 5; NA -> missing
    # 2-tree forest: x <= split goes left
    r = oe.rf_model([0, 3, 6], [2, 0, 0, 2, 0, 0], [3, 0, 0, 3, 0, 0], [-3, -1, -1, -3, -1, -1],
                    [1, 0, 0, 2, 0, 0], [1.0, 0, 0, 7.0, 0, 0], [0, 10.0, 20.0, 0, 1.0, 3.0])
    X = np.array([[1.0, 7.0], [1.5, 7.5]])
    assert np.allclose(oe.predict(r, X), [(10 + 1) / 2, (20 + 3) / 2])
    # 3-hinge earth
    e = oe.earth_model([1.0, 2.0, -1.0, 0.5], [[0, 0], [1, 0], [-1, 0], [0, 2]], [[0, 0], [3.0, 0], [3.0, 0], [0, 0]])
    X = np.array([[5.0, 4.0], [1.0, -2.0]])
    assert np.allclose(oe.predict(e, X), [1 + 2 * 2 - 0 + 0.5 * 4, 1 + 0 - 1 * 2 + 0.5 * -2])
    # 2-SV ksvm
    s = oe.svr_model([0.5, -0.25], [[0.0, 0.0], [1.0, 1.0]], 0.1, 0.5, [1.0, 2.0], [2.0, 4.0], 10.0, 3.0)
    x = np.array([[3.0, 6.0]])  # scaled: (1, 1)
    want = (0.5 * np.exp(-0.5 * 2) - 0.25 * 1.0 - 0.1) * 3.0 + 10.0
    assert np.allclose(oe.predict(s, x), want)
    # lm
    assert np.allclose(oe.predict(oe.lm_model([1.0, 2.0, 3.0]), np.array([[1.0, 1.0], [np.nan, 0]])), [6.0, np.nan], equal_nan=True)


def test_weight_selection_and_unrenormalised_sum():
    kept, wts, tot = oe.select_weights([0.31, 0.22, 0.004, 0.18, 0.27, 0.41])
    assert kept == "bgmrv" and wts == [0.31, 0.22, 0.18, 0.27, 0.41]
    assert np.isclose(tot, 1.394)  # the divisor keeps the dropped model's weight (V73:337,619)
    m1, m2 = oe.lm_model([1.0, 0.0]), oe.lm_model([3.0, 0.0])
    X = np.zeros((2, 1))
    assert np.allclose(oe.ensemble([m1, m2], [0.5, 0.25], 1.0, X), 0.5 * 1 + 0.25 * 3)


def test_stack_predictors_layer_order():
    cov = np.arange(2 * 3 * 4, dtype=np.float64).reshape(2, 3, 4)
    X = oe.stack_predictors(cov, (np.array([10.0, 11, 12, 13]), np.array([5.0, 4, 3])))
    assert X.shape == (12, 4)
    assert list(X[5]) == [5.0, 17.0, 11.0, 4.0]  # row 1, col 1: covs, LONG, LAT
