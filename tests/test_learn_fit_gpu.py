"""GPU: the learner fits of SURVEY.md 8f rank 4 through the C ABI -- mhs_svr_fit (kernlab::ksvm, V73:251/560),
mhs_nnet_fit (nnet::nnet, V73:249/463), mhs_gbm_staged_points + cv.gbm_step_search (machisplin.gbm.step's tree-count
search, V73:1765-1981) -- against oracle/fit.py (itself pinned to libsvm and to R's documented optim output in
tests/test_oracle_fit.py) and, for the SVR, against libsvm directly."""
import numpy as np
import pytest

from oracle import ensemble as oe
from oracle import fit as of

pytestmark = pytest.mark.gpu


def _data(n=300, p=5, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(n, p)) * np.array([1, 2, 3, 1, 5.0, 2, 1])[:p] + np.arange(p)
    y = np.sin(X[:, 0]) + 0.3 * X[:, 1] + 0.1 * rng.normal(size=n)
    return rng, X, y


@pytest.mark.parametrize("n,p,sigma", [(300, 5, 0.2), (1500, 3, 0.5), (97, 7, 0.05)])
def test_svr_fit_is_the_smo_of_the_oracle_and_of_libsvm(hip, n, p, sigma):
    sk = pytest.importorskip("sklearn.svm")
    rng, X, y = _data(n, p, seed=n)
    m = hip.models.Ksvm.fit(X, y, sigma)
    want, it = of.svr_fit(X, y, sigma)
    Z = (X - X.mean(0)) / X.std(0, ddof=1)
    t = (y - y.mean()) / y.std(ddof=1)
    K = of.rbf_gram(Z, sigma)
    assert of.svr_kkt_violation(K, t, m.beta) < 1e-3 * 1.001              # the stopping rule holds at what came back
    assert np.abs(m.beta).max() <= 1.0 and abs(m.beta.sum()) < 1e-10       # 0 <= alpha, alpha* <= C ; sum(alpha - alpha*) = 0
    # Two runs of the SMO that break a tie differently end at different points of the tol-flat set around the optimum:
    # at kernlab's tol = 0.001 the decision functions agree to a few tol, single coefficients to ~1e-2 ...
    ob_beta, ob_rho, _ = of.svr_smo(K, t)                                  # the oracle's beta on all rows
    assert np.abs(K @ (m.beta - ob_beta)).max() < 5e-3 and abs(m.params["b"] - ob_rho) < 5e-3
    assert 0.5 * it <= m.n_iter <= 2.0 * it
    ref = sk.SVR(kernel="rbf", gamma=sigma, C=1.0, epsilon=0.1, tol=1e-3).fit(Z, t)
    lib = np.zeros(n)
    lib[ref.support_] = ref.dual_coef_[0]
    assert np.abs(K @ (m.beta - lib)).max() < 5e-3 and abs(m.params["b"] + ref.intercept_[0]) < 5e-3
    # ... and with the stopping rule tightened to 1e-9 all three land on THE optimum of the dual
    tight = hip.models.Ksvm.fit(X, y, sigma, tol=1e-9)
    tb, trho, _ = of.svr_smo(K, t, tol=1e-9)
    assert np.abs(tight.beta - tb).max() < 1e-6 and abs(tight.params["b"] - trho) < 1e-7
    ref9 = sk.SVR(kernel="rbf", gamma=sigma, C=1.0, epsilon=0.1, tol=1e-9).fit(Z, t)
    lib9 = np.zeros(n)
    lib9[ref9.support_] = ref9.dual_coef_[0]
    # libsvm keeps Q in float32: its optimum is that of a slightly different Gram matrix
    assert np.abs(K @ (tight.beta - lib9)).max() < 5e-4 and abs(tight.params["b"] + ref9.intercept_[0]) < 5e-4
    np.testing.assert_allclose(m.params["x_center"], X.mean(0), rtol=1e-13)
    np.testing.assert_allclose(m.params["x_scale"], X.std(0, ddof=1), rtol=1e-13)
    # the fitted device model predicts like the oracle's evaluation of the same bundle, and like libsvm
    got = m.predict_points(X[:64])
    assert np.abs(got - oe.predict(m.params, X[:64])).max() < 1e-11 * np.abs(y).max()
    assert np.abs(got - (ref.predict(Z[:64]) * y.std(ddof=1) + y.mean())).max() < 5e-3 * y.std()


def test_svr_fit_with_more_stations_than_one_block_holds(hip):
    """n > 8 192: the SMO runs as a cooperative grid (two blocks here) with one slot exchange + barrier per reduction."""
    sk = pytest.importorskip("sklearn.svm")
    n, p, sigma = 8600, 3, 1.0
    rng = np.random.default_rng(5)
    X = rng.uniform(-2, 2, (n, p))
    y = np.sin(2 * X[:, 0]) * np.cos(X[:, 1]) + 0.2 * X[:, 2] + 0.3 * rng.normal(size=n)
    m = hip.models.Ksvm.fit(X, y, sigma)
    Z = (X - X.mean(0)) / X.std(0, ddof=1)
    t = (y - y.mean()) / y.std(ddof=1)
    K = of.rbf_gram(Z, sigma)
    assert of.svr_kkt_violation(K, t, m.beta) < 1e-3 * 1.001
    assert np.abs(m.beta).max() <= 1.0 and abs(m.beta.sum()) < 1e-9
    ref = sk.SVR(kernel="rbf", gamma=sigma, C=1.0, epsilon=0.1, tol=1e-3, cache_size=1000).fit(Z, t)
    lib = np.zeros(n)
    lib[ref.support_] = ref.dual_coef_[0]
    assert np.abs(K @ (m.beta - lib)).max() < 5e-3            # the two decision functions agree on every station
    assert abs(m.params["b"] + ref.intercept_[0]) < 5e-3
    again = hip.models.Ksvm.fit(X, y, sigma)
    assert np.array_equal(again.beta, m.beta) and again.n_iter == m.n_iter      # deterministic


def test_svr_fit_breaks_ties_toward_the_larger_index(hip):
    """libsvm's rule (the last of equal candidates wins, see tests/test_oracle_fit.py) on the device, within a block and
    across blocks: integer responses, ONE iteration (the C entry point reports 'no convergence' and still hands back
    beta), the two variables that moved are the oracle's; and to convergence the device follows the oracle's count."""
    import ctypes as C
    from machisplin_amd import _lib
    for n in (240, 8300):                                                # one block; two cooperative blocks
        rng, X, _ = _data(n=n, p=4, seed=3)
        y = rng.integers(180, 190, n).astype(float)
        Z = (X - X.mean(0)) / X.std(0, ddof=1)
        t = (y - y.mean()) / y.std(ddof=1)
        K = of.rbf_gram(Z, 0.3)
        want, _, _ = of.svr_smo(K, t, max_iter=1)
        Xf = np.asfortranarray(X)
        beta, xc, xs = np.zeros(n), np.empty(4), np.empty(4)
        b, yc, ys, it = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
        rc = _lib.lib().mhs_svr_fit(Xf.ctypes.data, y.ctypes.data, n, 4, 0.3, 1.0, 0.1, 1e-3, 1, beta.ctypes.data, C.byref(b),
                                    xc.ctypes.data, xs.ctypes.data, C.byref(yc), C.byref(ys), C.byref(it))
        assert rc != 0 and it.value == 1
        assert np.array_equal(np.flatnonzero(beta), np.flatnonzero(want)), (n, np.flatnonzero(beta), np.flatnonzero(want))
        top = np.flatnonzero(y == y.max())
        assert top[-1] in np.flatnonzero(beta)
        assert np.allclose(beta[np.flatnonzero(want)], want[np.flatnonzero(want)], rtol=1e-12)
    rng, X, _ = _data(n=240, p=4, seed=3)
    y = rng.integers(180, 190, 240).astype(float)
    m = hip.models.Ksvm.fit(X, y, 0.3)
    Z = (X - X.mean(0)) / X.std(0, ddof=1)
    ob, orho, oit = of.svr_smo(of.rbf_gram(Z, 0.3), (y - y.mean()) / y.std(ddof=1))
    assert abs(m.n_iter - oit) <= 0.05 * oit and np.abs(m.beta - ob).max() < 2e-2


@pytest.mark.parametrize("p", [3, 5, 7])
def test_nnet_fit_follows_vmmin(hip, p):
    rng, X, y = _data(400, p, seed=11 + p)
    H = 10
    w0 = rng.uniform(-0.7, 0.7, (p + 1) * H + H + 1)
    t = (y - y.min()) / (y - y.min()).max()
    # a short run: the device follows the oracle's iterates (same counts, same weights)
    short = hip.models.Nnet.fit(X, y, w0, maxit=25)
    w, val, nf, ng, fail = of.nnet_fit(X, t, w0, maxit=25)
    assert short.counts == (nf, ng) and short.fail == fail == 1
    assert np.abs(short.wts - w).max() < 1e-8 * np.abs(w).max() and abs(short.value - val) < 1e-10 * val
    # the whole fit (maxit = 10000 as V73:463): converged by vmmin's own test, no worse than the short run, and the
    # model object predicts with the fitted weights and the un-scaling of V73:469-470
    full = hip.models.Nnet.fit(X, y, w0)
    assert full.fail == 0 and full.value < short.value and full.counts[0] >= full.counts[1] > 25
    v, g = of.nnet_value_grad(full.wts, X, t, H)
    assert abs(v - full.value) < 1e-9 * v
    want = oe.predict_nnet(oe.nnet_model(full.wts, p, H, (y - y.min()).max(), y.min()), X[:50])
    assert np.abs(full.predict_points(X[:50]) - want).max() < 1e-12 * np.abs(y).max()
    again = hip.models.Nnet.fit(X, y, w0)
    assert np.array_equal(again.wts, full.wts) and again.counts == full.counts        # deterministic


def test_gbm_staged_predictions_and_the_tree_count_search(hip):
    from machisplin_amd import cv, synth
    rng, X, y = _data(600, 5, seed=3)
    X[rng.integers(0, 600, 12), rng.integers(0, 5, 12)] = np.nan                 # NA rows take the missing branch
    Xc = np.where(np.isnan(X), 0.0, X)
    selector = rng.integers(1, 4, 600)                                            # 3 folds
    folds, models = [], []
    for i in (1, 2, 3):
        tr = selector != i
        prm = synth.gbm_params(Xc[tr], y[tr], 100 + i, n_trees=2000, shrinkage=0.01)
        folds.append(prm)
        models.append(hip.models.from_param_dict(prm))
    step = 50
    P = models[0].staged_predict_points(X, step)
    want = of.gbm_staged_predictions(folds[0], X, step)
    assert P.shape == want.shape == (40, 600)
    assert np.abs(P - want).max() <= 1e-12 * np.abs(want).max()
    cut = dict(folds[0])                                                          # a model cut at 350 trees = stage 7
    k = 350
    cut["tree_offsets"] = folds[0]["tree_offsets"][:k + 1]
    for key in ("split_var", "split_val", "left", "right", "missing"):
        cut[key] = folds[0][key][:cut["tree_offsets"][-1]]
    # (the point path of mhs_predict_points adds the trees in several shares, so not bit for bit)
    assert np.abs(hip.models.from_param_dict(cut).predict_points(X) - P[6]).max() <= 1e-13 * np.abs(P[6]).max()
    got = cv.gbm_step_search(models, X, y, selector, step=step, max_trees=2000)
    ref = of.gbm_step_search(folds, X, y, selector, step=step, max_trees=2000)
    assert got is not None and ref is not None
    assert got[0] == ref[0] and np.array_equal(got[2], ref[2])
    assert np.abs(got[1] - ref[1]).max() < 1e-12 * ref[1].max()
    assert got[2][-1] <= 2000 and got[0] in got[2]
