"""CPU: the oracle's restatements of the learner FITS (oracle/fit.py) against what this
container can pin them to:
  * vmmin (R's optim "BFGS", the optimiser inside nnet::nnet) against the output R's own ?optim example prints for
    the Rosenbrock function -- value 9.594956e-18, counts 110 / 43;
  * the eps-SVR SMO against libsvm itself (scikit-learn's SVR);
  * the gbm.step stopping rule (V73:1872-1981) on crafted hold-out curves."""
import numpy as np
import pytest

from oracle import ensemble as oe
from oracle import fit as of


def test_vmmin_reproduces_the_rosenbrock_example_of_r_optim():
    """example(optim): optim(c(-1.2, 1), fr, grr, method = "BFGS") prints $value 9.594956e-18 and $counts 110 43
    (defaults maxit = 100, abstol = -Inf, reltol = sqrt(.Machine$double.eps))."""
    fr = lambda x: 100.0 * (x[1] - x[0] * x[0]) ** 2 + (1.0 - x[0]) ** 2
    grr = lambda x: np.array([-400.0 * x[0] * (x[1] - x[0] * x[0]) - 2.0 * (1.0 - x[0]), 200.0 * (x[1] - x[0] * x[0])])
    b, val, nf, ng, fail = of.vmmin([-1.2, 1.0], lambda x: (fr(x), grr(x)), maxit=100, abstol=-np.inf,
                                    reltol=np.sqrt(np.finfo(float).eps))
    assert (nf, ng, fail) == (110, 43, 0)
    assert abs(val - 9.594956e-18) < 2e-7 * 9.594956e-18   # a residual of 3e-9 squared: the last printed digit is rounding
    assert np.abs(b - 1.0).max() < 1e-8


def _data(n=300, p=5, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(n, p)) * np.array([1, 2, 3, 1, 5.0])[:p] + np.arange(p)
    y = np.sin(X[:, 0]) + 0.3 * X[:, 1] + 0.1 * rng.normal(size=n)
    return rng, X, y


def test_nnet_objective_gradient_and_fit():
    rng, X, y = _data()
    p, H = X.shape[1], 10
    t = (y - y.min()) / (y - y.min()).max()          # V73:455-459
    w0 = rng.uniform(-0.7, 0.7, (p + 1) * H + H + 1)
    v, g = of.nnet_value_grad(w0, X, t, H)
    assert abs(v - np.sum((oe.predict_nnet(oe.nnet_model(w0, p, H, 1.0, 0.0), X) - t) ** 2)) < 1e-12 * v   # same network as predict
    h = 1e-6
    for k in (0, 7, p + 1, (p + 1) * H, (p + 1) * H + 3):
        e = np.zeros_like(w0); e[k] = h
        fd = (of.nnet_value_grad(w0 + e, X, t, H)[0] - of.nnet_value_grad(w0 - e, X, t, H)[0]) / (2 * h)
        assert abs(fd - g[k]) < 1e-6 * max(1.0, abs(g[k]))
    w, val, nf, ng, fail = of.nnet_fit(X, t, w0, maxit=200)
    assert val < 0.2 * v and nf >= ng and fail in (0, 1)
    w2, val2, *_ = of.nnet_fit(X, t, w0, maxit=400)
    assert val2 <= val


def test_svr_smo_is_libsvm():
    sk = pytest.importorskip("sklearn.svm")
    rng, X, y = _data()
    sigma = 0.2
    Z = (X - X.mean(0)) / X.std(0, ddof=1)
    t = (y - y.mean()) / y.std(ddof=1)
    K = of.rbf_gram(Z, sigma)
    beta, rho, it = of.svr_smo(K, t)
    ref = sk.SVR(kernel="rbf", gamma=sigma, C=1.0, epsilon=0.1, tol=1e-3).fit(Z, t)
    want = np.zeros(t.size)
    want[ref.support_] = ref.dual_coef_[0]
    assert np.abs(beta - want).max() < 1e-4 and abs(rho + ref.intercept_[0]) < 1e-5
    assert of.svr_kkt_violation(K, t, beta) < 1e-3 and it > 100
    assert np.abs(beta).max() <= 1.0 and abs(beta.sum()) < 1e-12       # box and equality constraints
    m, _ = of.svr_fit(X, y, sigma)
    pred = oe.predict(m, X)
    assert np.abs(pred - (ref.predict(Z) * y.std(ddof=1) + y.mean())).max() < 1e-5 * y.std()
    assert m["alpha"].size == ref.support_.size


def test_svr_smo_breaks_ties_as_libsvm_does():
    """Integer responses (the bundled bio_1 column is 187, 224, 225 ...: data-raw/sampling.csv) tie the gradients at
    iteration 0, and libsvm's select_working_set scans t = 0 .. 2n-1 with '>=' / '<=': the LAST of equal candidates wins.
    After ONE iteration the two variables that moved are libsvm's (max_iter = 1), i being the last maximiser of y."""
    sk = pytest.importorskip("sklearn.svm")
    import warnings
    rng, X, _ = _data(n=240, p=4, seed=3)
    y = rng.integers(180, 190, 240).astype(float)                      # ten distinct values: 20-odd ties at the top
    Z = (X - X.mean(0)) / X.std(0, ddof=1)
    t = (y - y.mean()) / y.std(ddof=1)
    K = of.rbf_gram(Z, 0.3)
    beta, _, it = of.svr_smo(K, t, max_iter=1)
    moved = np.flatnonzero(beta)
    assert it == 1 and moved.size == 2
    top = np.flatnonzero(t == t.max())
    assert top.size > 5 and top[-1] in moved and not np.isin(top[:-1], moved).any()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = sk.SVR(kernel="rbf", gamma=0.3, C=1.0, epsilon=0.1, tol=1e-3, max_iter=1, shrinking=False).fit(Z, t)
    assert np.array_equal(np.sort(ref.support_), moved)
    assert np.allclose(ref.dual_coef_[0], beta[np.sort(ref.support_)], rtol=1e-5)
    # and to convergence the two agree as closely as on tie-free data (libsvm keeps Q in float32)
    full, rho, it_full = of.svr_smo(K, t)
    ref = sk.SVR(kernel="rbf", gamma=0.3, C=1.0, epsilon=0.1, tol=1e-3, shrinking=False).fit(Z, t)
    want = np.zeros(t.size)
    want[ref.support_] = ref.dual_coef_[0]
    assert np.abs(K @ (full - want)).max() < 5e-3
    if hasattr(ref, "n_iter_"):
        assert abs(int(np.ravel(ref.n_iter_)[0]) - it_full) <= 0.1 * it_full


def test_gbm_step_rule_on_crafted_curves():
    k = np.arange(1, 201)
    u_shaped = 1.0 + 0.5 * np.exp(-k / 10.0) + 1e-5 * (k - 60.0) ** 2 / 60.0      # minimum near stage 60
    target, cv, trees = of.gbm_step_rule([u_shaped, u_shaped + 0.01], 1e-6, 50, 10000)
    assert trees[0] == 50 and np.all(np.diff(trees) == 50) and cv.size == trees.size
    j = cv.size
    assert j >= 20 and np.mean(cv[j - 20:j - 9]) - np.mean(cv[j - 10:j]) <= 1e-6          # stopped by the rule ...
    assert np.mean(cv[j - 21:j - 10]) - np.mean(cv[j - 11:j - 1]) > 1e-6                  # ... at the first such stage
    assert target == trees[np.argmin(cv)] and 50 * 60 < target < trees[-1]                 # the minimum, past the quadratic's (stage 60)
    falling = 2.0 * np.exp(-k / 500.0)
    target, cv, trees = of.gbm_step_rule([falling], 1e-9, 50, 2000)
    assert trees[-1] == 2000 and target == 2000                                            # max.trees ends the loop
    rising = np.concatenate([[1.0, 0.9, 0.95], np.full(50, 0.8)])
    assert of.gbm_step_rule([rising], 1e-6, 50, 10000) is None                             # "restart with a smaller learning rate"
    with pytest.raises(ValueError):
        of.gbm_step_rule([falling[:30]], 1e-12, 50, 10000)
