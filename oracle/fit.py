"""Oracle: FITTING of the two learners SURVEY.md section 8(f) rank 4 names beside the linear member -- the RBF
support-vector regression (kernlab::ksvm, V73:251 in the CV loop, V73:560 final) and the one-hidden-layer network
(nnet::nnet(size=10, linout=TRUE, maxit=10000), V73:249, V73:463) -- and machisplin.gbm.step's tree-count search
(V73:1660-2239) over fold models that have already been grown.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED vs R: kernlab, nnet and gbm are CRAN
dependencies (DESCRIPTION:11, unpinned, not vendored); what follows restates their published algorithms:

  * ksvm(type="eps-svr", kernel="rbfdot", C=1, epsilon=0.1, tol=0.001, scaled=TRUE): kernlab's solver is the
    libsvm SMO (Fan, Chen, Lin 2005, "Working set selection using second order information"; libsvm's
    Solver::Solve / select_working_set / calculate_rho) on the 2n-variable dual of eps-SVR.  The dual optimum is
    unique (the RBF Gram matrix is positive definite), so any implementation of the same stopping rule agrees
    with kernlab's to the solver tolerance; tests cross-check against scikit-learn's SVR, which IS libsvm.
    kpar="automatic" (sigest on a random half of the rows) is RNG-dependent: sigma is an input here.
  * nnet: the objective is the sum over cases of (y - yhat)^2 (+ decay * sum w^2, decay = 0 at V73:463), minimised
    by R's optim(method="BFGS") = `vmmin` (R sources src/appl/optim.c) with abstol = 1e-4, reltol = 1e-8 from
    the initial weights runif(-0.7, 0.7) (RNG-dependent: the initial weights are an input here).
  * gbm.step: hold-out deviance of every fold model at n.trees = 50, 100, ... (gaussian: mean squared error,
    machisplin.calc.deviance V73:2250-2285), the stopping rule of V73:1884-1961 and the choice of the tree count at
    the minimum of the fold-mean curve (V73:1976-1981).
"""
from __future__ import annotations

import numpy as np

from . import ensemble as oe


# ------------------------------------------------------------------------------------------------ eps-SVR (SMO) --
def rbf_gram(Z, sigma):
    """kernlab rbfdot: K_ij = exp(-sigma |z_i - z_j|^2)."""
    sq = (Z * Z).sum(1)
    d2 = np.maximum(sq[:, None] + sq[None, :] - 2.0 * (Z @ Z.T), 0.0)
    np.fill_diagonal(d2, 0.0)
    return np.exp(-sigma * d2)


def svr_smo(K, y, C=1.0, epsilon=0.1, tol=1e-3, max_iter=10_000_000):
    """libsvm Solver::Solve for eps-SVR without shrinking.  Variables a[0:n] = alpha, a[n:2n] = alpha*; signs
    s = (+1, -1); linear term p = (eps - y, eps + y); Q_tu = s_t s_u K_{t mod n, u mod n}.
    Returns beta = alpha - alpha*, rho (decision = K beta - rho) and the iteration count."""
    n = y.size
    TAU = 1e-12
    a = np.zeros(2 * n)
    s = np.concatenate([np.ones(n), -np.ones(n)])
    p = np.concatenate([epsilon - y, epsilon + y])
    G = p.copy()                                      # gradient of 1/2 a'Qa + p'a at a = 0
    QD = np.concatenate([np.diag(K), np.diag(K)])
    it = 0
    while it < max_iter:
        up = ((s > 0) & (a < C)) | ((s < 0) & (a > 0))
        low = ((s > 0) & (a > 0)) | ((s < 0) & (a < C))
        mg = -s * G
        if not up.any() or not low.any():
            break
        iu = np.flatnonzero(up)                          # libsvm scans t = 0 .. 2n-1 with '>=': the LAST maximiser wins a tie
        i = int(iu[iu.size - 1 - np.argmax(mg[iu][::-1])])
        gmax = mg[i]
        gmax2 = np.max(-mg[low])
        if gmax + gmax2 < tol:
            break
        Qi = s[i] * s * np.concatenate([K[i % n], K[i % n]])
        cand = low & (mg < gmax)
        b = gmax - mg
        aq = QD[i] + QD - 2.0 * s[i] * s * Qi
        aq = np.where(aq > 0, aq, TAU)
        obj = np.where(cand, -(b * b) / aq, np.inf)
        j = int(obj.size - 1 - np.argmin(obj[::-1]))     # 'obj_diff <= obj_diff_min': the last minimiser
        if not np.isfinite(obj[j]):
            break
        Qj = s[j] * s * np.concatenate([K[j % n], K[j % n]])
        oi, oj = a[i], a[j]
        if s[i] != s[j]:
            quad = QD[i] + QD[j] + 2.0 * Qi[j]
            if quad <= 0:
                quad = TAU
            delta = (-G[i] - G[j]) / quad
            diff = a[i] - a[j]
            a[i] += delta
            a[j] += delta
            if diff > 0:
                if a[j] < 0:
                    a[j] = 0.0
                    a[i] = diff
            else:
                if a[i] < 0:
                    a[i] = 0.0
                    a[j] = -diff
            if diff > 0:                              # C_i - C_j = 0
                if a[i] > C:
                    a[i] = C
                    a[j] = C - diff
            else:
                if a[j] > C:
                    a[j] = C
                    a[i] = C + diff
        else:
            quad = QD[i] + QD[j] - 2.0 * Qi[j]
            if quad <= 0:
                quad = TAU
            delta = (G[i] - G[j]) / quad
            ssum = a[i] + a[j]
            a[i] -= delta
            a[j] += delta
            if ssum > C:
                if a[i] > C:
                    a[i] = C
                    a[j] = ssum - C
            else:
                if a[j] < 0:
                    a[j] = 0.0
                    a[i] = ssum
            if ssum > C:
                if a[j] > C:
                    a[j] = C
                    a[i] = ssum - C
            else:
                if a[i] < 0:
                    a[i] = 0.0
                    a[j] = ssum
        G += Qi * (a[i] - oi) + Qj * (a[j] - oj)
        it += 1
    # calculate_rho
    sg = s * G
    free = (a > 0) & (a < C)
    if free.any():
        rho = sg[free].sum() / free.sum()
    else:
        ub_set = ((a >= C) & (s < 0)) | ((a <= 0) & (s > 0))
        lb_set = ((a >= C) & (s > 0)) | ((a <= 0) & (s < 0))
        ub = sg[ub_set].min() if ub_set.any() else np.inf
        lb = sg[lb_set].max() if lb_set.any() else -np.inf
        rho = 0.5 * (ub + lb)
    return a[:n] - a[n:], float(rho), it


def svr_kkt_violation(K, y, beta, C=1.0, epsilon=0.1):
    """The stopping quantity of the SMO at `beta` (m(a) - M(a) of Fan et al.): <= tol at a solution."""
    n = y.size
    a = np.concatenate([np.maximum(beta, 0.0), np.maximum(-beta, 0.0)])
    s = np.concatenate([np.ones(n), -np.ones(n)])
    Kb = K @ beta
    G = np.concatenate([Kb + epsilon - y, -Kb + epsilon + y])
    up = ((s > 0) & (a < C)) | ((s < 0) & (a > 0))
    low = ((s > 0) & (a > 0)) | ((s < 0) & (a < C))
    return float(np.max((-s * G)[up]) + np.max((s * G)[low]))


def svr_fit(X, y, sigma, C=1.0, epsilon=0.1, tol=1e-3):
    """kernlab::ksvm(mod.form, data) (V73:251, V73:560) for a numeric response: x and y scaled to zero mean and unit
    standard deviation (scaled = TRUE), eps-svr on the scaled data; returns the parameter bundle of
    oracle.ensemble.svr_model (support vectors = rows with beta != 0) and the iteration count."""
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    xc, xs = X.mean(0), X.std(0, ddof=1)
    yc, ys = y.mean(), y.std(ddof=1)
    Z = (X - xc) / xs
    t = (y - yc) / ys
    K = rbf_gram(Z, sigma)
    beta, rho, it = svr_smo(K, t, C, epsilon, tol)
    sv = np.flatnonzero(beta != 0.0)
    return oe.svr_model(beta[sv], Z[sv], rho, sigma, xc, xs, yc, ys), it


# ------------------------------------------------------------------------------------------- nnet (BFGS, vmmin) --
def nnet_value_grad(w, X, y, size):
    """nnet's objective for linout = TRUE, decay = 0: sum over cases of (y - yhat)^2, and its gradient, weights in
    nnet order (per hidden unit: bias, p inputs; then output bias, `size` hidden->output weights)."""
    n, p = X.shape
    W1 = w[:(p + 1) * size].reshape(size, p + 1)
    w2 = w[(p + 1) * size:]
    z = W1[:, 0][None, :] + X @ W1[:, 1:].T
    h = oe._nnet_sigmoid(z)
    out = w2[0] + h @ w2[1:]
    err = out - y
    val = float(np.sum(err * err))
    d_out = 2.0 * err
    g2 = np.concatenate([[d_out.sum()], h.T @ d_out])
    dz = (d_out[:, None] * w2[1:][None, :]) * h * (1.0 - h)
    g1 = np.column_stack([dz.sum(0), dz.T @ X])
    return val, np.concatenate([g1.ravel(), g2])


def vmmin(b0, fn_gr, maxit=10000, abstol=1e-4, reltol=1e-8):
    """R's optim(method = "BFGS"): variable-metric minimiser of J. C. Nash as coded in src/appl/optim.c (vmmin).
    fn_gr(b) -> (value, gradient).  Returns b, value, function count, gradient count, fail flag."""
    stepredn, acctol, reltest = 0.2, 1e-4, 10.0
    b = np.array(b0, dtype=np.float64)
    n = b.size
    if maxit <= 0:
        return b, fn_gr(b)[0], 0, 0, 0
    f, g = fn_gr(b)
    if not np.isfinite(f):
        raise ValueError("initial value in 'vmmin' is not finite")
    fmin = f
    funcount = gradcount = 1
    it = 1
    ilast = gradcount
    B = np.eye(n)
    while True:
        if ilast == gradcount:
            B = np.eye(n)
        X = b.copy()
        c = g.copy()
        t = -(B @ g)
        gradproj = float(t @ g)
        if gradproj < 0.0:
            steplength = 1.0
            accpoint = False
            while True:
                b = X + steplength * t
                count = int(np.sum(reltest + X == reltest + b))
                if count < n:
                    f, gnew = fn_gr(b)
                    funcount += 1
                    accpoint = bool(np.isfinite(f) and f <= fmin + gradproj * steplength * acctol)
                    if not accpoint:
                        steplength *= stepredn
                if count == n or accpoint:
                    break
            enough = (f > abstol) and abs(f - fmin) > reltol * (abs(fmin) + reltol)
            if not enough:
                count = n
                fmin = f
            if count < n:
                fmin = f
                g = gnew
                gradcount += 1
                it += 1
                t = steplength * t
                c = g - c
                D1 = float(t @ c)
                if D1 > 0:
                    Xc = B @ c
                    D2 = 1.0 + float(Xc @ c) / D1
                    B = B + (D2 * np.outer(t, t) - np.outer(Xc, t) - np.outer(t, Xc)) / D1
                else:
                    ilast = gradcount
            else:
                if ilast < gradcount:
                    count = 0
                    ilast = gradcount
        else:
            count = 0
            if ilast == gradcount:
                count = n
            else:
                ilast = gradcount
        if it >= maxit:
            break
        if gradcount - ilast > 2 * n:
            ilast = gradcount
        if count == n and ilast == gradcount:
            break
    return b, fmin, funcount, gradcount, int(it >= maxit)


def nnet_fit(X, y, wts0, size=10, maxit=10000, abstol=1e-4, reltol=1e-8):
    """nnet::nnet(mod.form, data = trainNN, size = 10, linout = TRUE, maxit = 10000) (V73:249, V73:463) from the
    initial weights wts0.  The caller scales the response as V73:455-459 does."""
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    w, val, nf, ng, fail = vmmin(wts0, lambda w: nnet_value_grad(w, X, y, size), maxit, abstol, reltol)
    return w, val, nf, ng, fail


# ------------------------------------------------------------------------------------- gbm.step tree-count search --
def gaussian_deviance(obs, pred, calc_mean=True):
    """machisplin.calc.deviance(family = "gaussian") (V73:2250-2285): sum (obs - pred)^2 -- the weights argument is
    not used by this family -- divided by the number of observations when calc.mean."""
    obs = np.asarray(obs, dtype=np.float64)
    d = float(np.sum((obs - pred) * (obs - pred)))
    return d / obs.size if calc_mean else d


def gbm_staged_predictions(m, X, step):
    """predict.gbm(model, X, n.trees = step, 2 step, ...) for one gbm parameter bundle: (n_stages, rows)."""
    off = m["tree_offsets"]
    nt = len(off) - 1
    out = []
    for k in range(step, nt + 1, step):
        sub = dict(m)
        sub["tree_offsets"] = off[:k + 1]
        out.append(oe.predict_gbm(sub, X))
    return np.array(out)


def gbm_step_search(fold_models, X, y, selector, step=50, tolerance=0.001, max_trees=10000, site_weights=None):
    """The tree-count search of machisplin.gbm.step (V73:1765-1990) over fold models that have been grown far enough
    (the growing itself, gbm::gbm / gbm.more with bag.fraction = 0.5, is RNG-dependent and stays in the package):
    fold i's model predicts its hold-out rows (selector == i) at n.trees = step, 2 step, ... (V73:1843, 1919);
    cv.loss.values[j] = mean over the folds of the hold-out deviance (V73:1866, 1942-1946); stages are added while
    delta.deviance > tolerance.test and n.fitted < max.trees (V73:1884), where from the 20th stage on delta.deviance =
    mean(cv[j-19 .. j-9]) - mean(cv[j-9 .. j]) (V73:1957-1961); a loss that rises within the first four stages
    aborts ("restart with a smaller learning rate", V73:1948-1955: returns None); the tree count is the first
    stage with the smallest loss (V73:1976-1981).  Returns (target_trees, cv_loss_values, trees_fitted)."""
    y = np.asarray(y, dtype=np.float64)
    w = np.ones_like(y) if site_weights is None else np.asarray(site_weights, dtype=np.float64)
    u = np.full_like(y, np.sum(y * w) / np.sum(w))
    tolerance_test = gaussian_deviance(y, u, calc_mean=False) / y.size * tolerance      # V73:1786-1794
    staged = []
    for i, m in enumerate(fold_models):
        mask = selector == i + 1
        P = gbm_staged_predictions(m, X[mask], step)
        staged.append(np.array([gaussian_deviance(y[mask], P[k]) for k in range(P.shape[0])]))
    return gbm_step_rule(staged, tolerance_test, step, max_trees)


def gbm_step_rule(staged, tolerance_test, step, max_trees):
    """The loop of V73:1872-1981 on the folds' hold-out deviance curves staged[i][k] (k-th stage = (k + 1) step trees)."""
    n_fitted = step
    trees = [n_fitted]
    cv = [float(np.mean([s[0] for s in staged]))]
    delta = 1.0
    j = 1
    while delta > tolerance_test and n_fitted < max_trees:
        n_fitted += step
        trees.append(n_fitted)
        j += 1
        if j > len(staged[0]):
            raise ValueError("fold models have fewer trees than the search needs")
        cv.append(float(np.mean([s[j - 1] for s in staged])))
        if j < 5 and cv[j - 1] > cv[j - 2]:
            return None
        if j >= 20:
            test1 = np.mean(cv[j - 10:j])
            test2 = np.mean(cv[j - 20:j - 9])
            delta = test2 - test1
    cv = np.array(cv)
    return trees[int(np.argmax(cv == cv.min()))], cv, np.array(trees)
