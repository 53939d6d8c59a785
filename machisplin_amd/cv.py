"""Step 1's data-parallel part: hold-out predictions of the six learners for the
k-fold cross-validation (V73:225-319) on the GPU, and the ensemble weight search that consumes them
(V73:326-393); rank 4: the tree-count search of machisplin.gbm.step over grown fold models (gbm_step_search).

Fitting the fold models stays in the CRAN packages (R); what runs here is what R does with
``terra::predict(model, test)`` inside the fold loop: every fold's models evaluated at that fold's hold-out
rows through ``mhs_predict_points``, the residual vectors concatenated in fold order, and
``optimx(par = 0.5, lower = 0, upper = 1, method = "L-BFGS-B")`` on

    fit(k) = sum_i ( sum_m k_m r_{i,m} / sum_m k_m )^2 = k' (R'R) k / (1'k)^2 .

The objective is scale-invariant, so the optimiser's end point on the minimising ray decides the rounded
weights (V73:340-362); this module uses SciPy's L-BFGS-B (the same Nocedal/Zhu code base R's optim wraps)
with the same start, bounds and objective -- the iterate sequence of R's build is NOT reproduced bit for
bit, and R-side integration keeps optimx in R."""
from __future__ import annotations

import numpy as np

from .models import Model, select_weights

ORDER_ALL = "bgnmrv"      # OptX$p1..p6: brt, gam, nn, mars, rf, svm (V73:326-331)
ORDER_SMOOTH = "gnmv"     # smooth.outputs.only = TRUE (V73:366-372)


def holdout_rows(kfolds, v: int, n_rows: int):
    """V73:228-232: with more than 4000 rows the model is TRAINED on fold v and tested on the other nine."""
    kfolds = np.asarray(kfolds)
    return np.flatnonzero(kfolds != v) if n_rows > 4000 else np.flatnonzero(kfolds == v)


def train_rows(kfolds, v: int, n_rows: int):
    """The complement of :func:`holdout_rows` (V73:228-232): fold v itself when there are more than 4000 rows."""
    kfolds = np.asarray(kfolds)
    return np.flatnonzero(kfolds == v) if n_rows > 4000 else np.flatnonzero(kfolds != v)


def fit_linear_folds(X, resp, kfolds):
    """``mod.gam.tps.elev <- mgcv::gam(mod.form, data = train)`` for every fold (V73:252) on the device: the one
    member whose fit is deterministic (least squares, :meth:`models.Gam.fit`).  Returns the fold models in fold
    order, ready for the ``g`` slot of ``fold_models`` in :func:`cv_residuals`."""
    from .models import Gam
    X = np.asarray(X, dtype=np.float64)
    resp = np.asarray(resp, dtype=np.float64)
    out = []
    for v in range(1, int(np.max(kfolds)) + 1):
        tr = train_rows(kfolds, v, X.shape[0])
        out.append(Gam.fit(X[tr], resp[tr]))
    return out


def cv_residuals(fold_models, X, resp, kfolds, labels: str = ORDER_ALL):
    """mfit.<model>.full of V73:258-319: for fold v = 1..nfolds, ``test$resp - predict(model_v, test)`` on the
    hold-out rows, concatenated in fold order.  ``fold_models[v-1]`` maps a label in ``labels`` to the device
    model (:class:`machisplin_amd.models.Model`) fitted on fold v's training rows; ``X`` is the n x p predictor
    matrix (covariates, LONG, LAT), ``kfolds`` the 1-based fold label of every row.
    Returns an (n_holdout_total, len(labels)) float64 matrix, columns in ``labels`` order."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    resp = np.asarray(resp, dtype=np.float64)
    cols = {lab: [] for lab in labels}
    for v, models in enumerate(fold_models, start=1):
        rows = holdout_rows(kfolds, v, X.shape[0])
        Xt = np.ascontiguousarray(X[rows])
        for lab in labels:
            m = models[lab]
            if not isinstance(m, Model):
                raise TypeError("fold %d: model %r is not a device model" % (v, lab))
            cols[lab].append(resp[rows] - m.predict_points(Xt))
    return np.column_stack([np.concatenate(cols[lab]) for lab in labels])


def optx_objective(k, gram):
    """machisplin.optimx.internal (V73:329-331, 369-371) through the Gram matrix of the residual columns."""
    k = np.asarray(k, dtype=np.float64)
    s = k.sum()
    return float(k @ gram @ k) / (s * s)


def optx_weights(residuals, smooth_only: bool = False):
    """OptX of V73:333 / 373: minimise the objective from par = 0.5 in [0, 1]^m with L-BFGS-B (numerical
    gradient, as optimx does without ``gr``), then apply V73:336-362.  Returns (p, kept labels, kept rounded
    weights, unrounded total)."""
    from scipy.optimize import minimize
    R = np.asarray(residuals, dtype=np.float64)
    labels = ORDER_SMOOTH if smooth_only else ORDER_ALL
    if R.ndim != 2 or R.shape[1] != len(labels):
        raise ValueError("residuals must have %d columns (%s)" % (len(labels), labels))
    gram = R.T @ R
    res = minimize(optx_objective, np.full(len(labels), 0.5), args=(gram,), method="L-BFGS-B",
                   bounds=[(0.0, 1.0)] * len(labels))
    kept, wts, tot = select_weights(res.x, labels)
    return res.x, kept, wts, tot


def gbm_step_search(fold_models, X, y, selector, step: int = 50, tolerance: float = 0.001, max_trees: int = 10000,
                    site_weights=None):
    """The tree-count search of ``machisplin.gbm.step`` (V73:1765-1981) over fold models that gbm has grown far enough
    (growing them -- gbm::gbm / gbm.more with bag.fraction = 0.5 -- is RNG-dependent and stays in the package):

    * fold i's model predicts its hold-out rows (``selector == i``) at n.trees = step, 2 step, ... in ONE device walk
      (:meth:`models.Gbm.staged_predict_points`; R calls predict.gbm once per stage, V73:1843, 1919);
    * ``cv.loss.values[j]`` = mean over the folds of the hold-out deviance, gaussian = mean squared error
      (machisplin.calc.deviance, V73:2250-2285; V73:1866, 1942-1946);
    * stages are added while ``delta.deviance > tolerance.test`` and ``n.fitted < max.trees`` (V73:1884); from the
      20th stage on ``delta.deviance = mean(cv[j-19 .. j-9]) - mean(cv[j-9 .. j])`` (V73:1957-1961);
      ``tolerance.test`` = tolerance x the mean total deviance (tolerance.method = "auto", V73:1786-1794);
    * a loss that rises within the first four stages aborts (V73:1948-1955: returns None, R prints "restart model
      with a smaller learning rate");
    * the tree count is the first stage with the smallest loss (V73:1976-1981).

    Returns ``(target_trees, cv_loss_values, trees_fitted)``."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    selector = np.asarray(selector)
    w = np.ones_like(y) if site_weights is None else np.asarray(site_weights, dtype=np.float64)
    u = np.sum(y * w) / np.sum(w)
    tolerance_test = float(np.sum((y - u) * (y - u))) / y.size * tolerance
    staged = []
    for i, m in enumerate(fold_models):
        mask = selector == i + 1
        P = m.staged_predict_points(X[mask], step)
        d = y[mask][None, :] - P
        staged.append(np.sum(d * d, axis=1) / int(mask.sum()))
    n_fitted = step
    trees = [n_fitted]
    cv = [float(np.mean([s[0] for s in staged]))]
    delta, j = 1.0, 1
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
            delta = float(np.mean(cv[j - 20:j - 9]) - np.mean(cv[j - 10:j]))
    cv = np.array(cv)
    return trees[int(np.argmax(cv == cv.min()))], cv, np.array(trees)
