"""Oracle: per-cell evaluation of the six ensemble members and their weighted sum, as
``terra::predict(rast_stack, model)`` reaches the CRAN packages' predict methods.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED vs R: gbm,
randomForest, nnet, earth, kernlab and mgcv are CRAN dependencies (DESCRIPTION:11,
unpinned, not vendored); what follows restates their published prediction algorithms
 and is cross-checked against scikit-learn
evaluators of the same structures in tests/.

Reference call sites (V73):
  gbm     terra::predict(rast_stack, brt, n.trees=best.trees, type="response")  V73:497,499
  rf      terra::predict(rast_stack, rf, type="response", ...)                  V73:521,523
  nnet    terra::predict(rast_stack, nn) * max2.resp.f + min.resp.f             V73:468-475
  earth   terra::predict(rast_stack, mars)                                      V73:543,545
  ksvm    terra::predict(rast_stack, svm, na.rm=TRUE)                           V73:582,584
  gam     terra::predict(rast_stack, gam)   (no s() terms => linear model)      V73:604,606
  sum     pred.elev + pred.elev.2 * wt ; pred.elev / OptX.mfit.wt.tot           V73:471..605, 619

X is (cells, p) float64 with the predictors in rast_stack layer order: the C covariates,
then LONG, then LAT (V73:138).  NaN is NA.
"""
from __future__ import annotations

import numpy as np


# ------------------------------------------------------------------ parameter bundles --
def lm_model(coef):
    """mgcv::gam with a purely parametric formula (V73:195,600): coefficients[p+1],
    intercept first."""
    return {"kind": "lm", "coef": np.asarray(coef, dtype=np.float64)}


def lm_fit(X, y):
    """mgcv::gam(mod.form, data) with the purely parametric formula of V73:195 (V73:252 in the CV loop, V73:600
    for the final model) = ordinary least squares on [1 X]; returns coefficients[p+1], intercept first."""
    X = np.asarray(X, dtype=np.float64)
    A = np.column_stack([np.ones(X.shape[0]), X])
    coef, *_ = np.linalg.lstsq(A, np.asarray(y, dtype=np.float64), rcond=None)
    return coef


def nnet_model(wts, p, size, y_scale, y_shift):
    """nnet(size=10, linout=TRUE) (V73:463).  wts in nnet order: for each hidden unit its
    bias then its p input weights; then the output bias and the `size` hidden->output
    weights.  y_scale/y_shift = max2.resp.f / min.resp.f (V73:455-459,469-470)."""
    wts = np.asarray(wts, dtype=np.float64)
    assert wts.size == (p + 1) * size + size + 1
    return {"kind": "nnet", "wts": wts, "p": p, "size": size, "y_scale": float(y_scale),
            "y_shift": float(y_shift)}


def earth_model(coef, dirs, cuts):
    """earth(degree=1) (V73:539): selected terms only.  dirs[k, v] in {0, 1, -1, 2}:
    0 unused, 1 max(0, x - cut), -1 max(0, cut - x), 2 linear x; a term is the product of
    its factors (the intercept term has no factor)."""
    dirs = np.asarray(dirs, dtype=np.int32)
    cuts = np.asarray(cuts, dtype=np.float64)
    coef = np.asarray(coef, dtype=np.float64)
    assert dirs.shape == cuts.shape and dirs.shape[0] == coef.size
    return {"kind": "earth", "coef": coef, "dirs": dirs, "cuts": cuts}


def svr_model(alpha, sv, b, sigma, x_center, x_scale, y_center, y_scale):
    """kernlab::ksvm eps-svr, rbfdot, scaled=TRUE (V73:560): alpha[nSV] signed
    coefficients, sv[nSV, p] support vectors in SCALED coordinates."""
    return {"kind": "svr", "alpha": np.asarray(alpha, dtype=np.float64),
            "sv": np.asarray(sv, dtype=np.float64), "b": float(b), "sigma": float(sigma),
            "x_center": np.asarray(x_center, dtype=np.float64),
            "x_scale": np.asarray(x_scale, dtype=np.float64),
            "y_center": float(y_center), "y_scale": float(y_scale)}


def gbm_model(init_f, tree_offsets, split_var, split_val, left, right, missing):
    """gbm object restricted to n.trees = best.trees (V73:497): concatenated per-tree
    node arrays (gbm's SplitVar [0-based, -1 = terminal], SplitCodePred, LeftNode,
    RightNode, MissingNode; child indices are tree-local), tree t owning nodes
    tree_offsets[t]:tree_offsets[t+1].  Terminal SplitCodePred already carries the
    shrinkage."""
    return {"kind": "gbm", "init_f": float(init_f),
            "tree_offsets": np.asarray(tree_offsets, dtype=np.int64),
            "split_var": np.asarray(split_var, dtype=np.int32),
            "split_val": np.asarray(split_val, dtype=np.float64),
            "left": np.asarray(left, dtype=np.int32), "right": np.asarray(right, dtype=np.int32),
            "missing": np.asarray(missing, dtype=np.int32)}


def rf_model(tree_offsets, left, right, status, best_var, split, node_pred):
    """randomForest regression forest (V73:517): concatenated per-tree columns of
    $forest (leftDaughter/rightDaughter 1-based tree-local, 0 at terminals; nodestatus
    -1 terminal / -3 split; bestvar 1-based; xbestsplit; nodepred)."""
    return {"kind": "rf", "tree_offsets": np.asarray(tree_offsets, dtype=np.int64),
            "left": np.asarray(left, dtype=np.int32), "right": np.asarray(right, dtype=np.int32),
            "status": np.asarray(status, dtype=np.int32),
            "best_var": np.asarray(best_var, dtype=np.int32),
            "split": np.asarray(split, dtype=np.float64),
            "node_pred": np.asarray(node_pred, dtype=np.float64)}


# ---------------------------------------------------------------------- evaluators --
def _na_rows(X):
    return np.isnan(X).any(axis=1)


def predict_lm(m, X):
    c = m["coef"]
    out = np.full(X.shape[0], c[0])
    for j in range(X.shape[1]):  # same left-to-right order as the kernel
        out = out + c[j + 1] * X[:, j]
    return out  # NaN in any predictor propagates


def _nnet_sigmoid(z):
    """nnet.c: sigmoid() saturates exactly outside [-15, 15]."""
    with np.errstate(over="ignore"):
        s = 1.0 / (1.0 + np.exp(-z))
    return np.where(z < -15.0, 0.0, np.where(z > 15.0, 1.0, s))


def predict_nnet(m, X):
    p, H, w = m["p"], m["size"], m["wts"]
    na = _na_rows(X)
    Xs = np.where(na[:, None], 0.0, X)
    out = np.full(X.shape[0], w[(p + 1) * H])
    for h in range(H):
        base = h * (p + 1)
        z = np.full(X.shape[0], w[base])
        for j in range(p):
            z = z + w[base + 1 + j] * Xs[:, j]
        out = out + w[(p + 1) * H + 1 + h] * _nnet_sigmoid(z)
    out = out * m["y_scale"] + m["y_shift"]
    return np.where(na, np.nan, out)


def predict_earth(m, X):
    na = _na_rows(X)
    out = np.zeros(X.shape[0])
    for k in range(m["coef"].size):
        term = np.ones(X.shape[0])
        for v in range(X.shape[1]):
            d = m["dirs"][k, v]
            if d == 0:
                continue
            if d == 2:
                f = X[:, v]
            elif d == 1:
                f = np.maximum(0.0, X[:, v] - m["cuts"][k, v])
            else:
                f = np.maximum(0.0, m["cuts"][k, v] - X[:, v])
            term = term * f
        out = out + m["coef"][k] * term
    return np.where(na, np.nan, out)


def predict_svr(m, X, block=2048):
    na = _na_rows(X)
    Xs = (np.where(na[:, None], 0.0, X) - m["x_center"]) / m["x_scale"]
    sv, alpha, sigma = m["sv"], m["alpha"], m["sigma"]
    out = np.empty(X.shape[0])
    for s in range(0, X.shape[0], block):
        xb = Xs[s:s + block]
        d2 = ((xb[:, None, :] - sv[None, :, :]) ** 2).sum(axis=2)
        out[s:s + block] = np.exp(-sigma * d2) @ alpha
    out = (out - m["b"]) * m["y_scale"] + m["y_center"]
    return np.where(na, np.nan, out)


def predict_gbm(m, X):
    """gbm_pred: NA -> MissingNode, x < split -> LeftNode else RightNode."""
    n = X.shape[0]
    out = np.full(n, m["init_f"])
    off = m["tree_offsets"]
    rows = np.arange(n)
    for t in range(off.size - 1):
        o = off[t]
        node = np.zeros(n, dtype=np.int64)
        var = m["split_var"][o + node]
        active = var >= 0
        while active.any():
            idx = o + node[active]
            v = m["split_var"][idx]
            x = X[rows[active], v]
            nxt = np.where(np.isnan(x), m["missing"][idx],
                           np.where(x < m["split_val"][idx], m["left"][idx], m["right"][idx]))
            node[active] = nxt
            var = m["split_var"][o + node]
            active = var >= 0
        out = out + m["split_val"][o + node]
    return out


def predict_rf(m, X):
    """regForest/predictRegTree: x <= xbestsplit -> left daughter; mean over trees."""
    n = X.shape[0]
    na = _na_rows(X)
    Xs = np.where(na[:, None], 0.0, X)
    off = m["tree_offsets"]
    ntree = off.size - 1
    acc = np.zeros(n)
    rows = np.arange(n)
    for t in range(ntree):
        o = off[t]
        node = np.zeros(n, dtype=np.int64)
        active = m["status"][o + node] != -1
        while active.any():
            idx = o + node[active]
            x = Xs[rows[active], m["best_var"][idx] - 1]
            node[active] = np.where(x <= m["split"][idx], m["left"][idx], m["right"][idx]) - 1
            active = m["status"][o + node] != -1
        acc = acc + m["node_pred"][o + node]
    return np.where(na, np.nan, acc / ntree)


PREDICTORS = {"lm": predict_lm, "nnet": predict_nnet, "earth": predict_earth, "svr": predict_svr,
              "gbm": predict_gbm, "rf": predict_rf}


def predict(model, X):
    return PREDICTORS[model["kind"]](model, np.asarray(X, dtype=np.float64))


def ensemble(models, weights, wt_total, X):
    """V73:447-619: pred = (((p1*w1) + p2*w2) + ...) / OptX.mfit.wt.tot, in mods.run order;
    `weights` are the rounded kept weights, `wt_total` the UNROUNDED total over all
    candidates (V73:337-338,376-377).  NA propagates."""
    acc = None
    for m, w in zip(models, weights):
        pk = predict(m, X) * w
        acc = pk if acc is None else acc + pk
    return acc / wt_total


def select_weights(p_opt, labels="bgnmrv"):
    """V73:336-362 / 375-392: keep model k iff round(p_k, 2) > 0.05 * sum(p); kept weight is
    round(p_k, 2); the divisor is the unrounded sum over ALL candidates."""
    p_opt = np.asarray(p_opt, dtype=np.float64)
    tot = float(p_opt.sum())
    cut = 0.05 * tot
    kept, wts = "", []
    for lab, pk in zip(labels, p_opt):
        r = float(np.round(pk, 2))
        if r > cut:
            kept += lab
            wts.append(r)
    return kept, wts, tot


def stack_predictors(covars, geom_xy):
    """rast_stack <- c(covar.ras, LONG, LAT) (V73:138): covars (C, nrow, ncol) planes,
    geom_xy = (x[ncol], y[nrow]) cell-centre coordinates -> X (cells, C+2)."""
    x, y = geom_xy
    C, nrow, ncol = covars.shape
    X = np.empty((nrow * ncol, C + 2))
    for k in range(C):
        X[:, k] = covars[k].reshape(-1)
    X[:, C] = np.tile(x, nrow)
    X[:, C + 1] = np.repeat(y, ncol)
    return X


# ------------------------------------------------- Step 1: CV residuals + weight search --
def holdout_rows(kfolds, v, n_rows):
    """V73:228-232: train on fold v / test on the rest when the table has more than 4000 rows."""
    kfolds = np.asarray(kfolds)
    return np.flatnonzero(kfolds != v) if n_rows > 4000 else np.flatnonzero(kfolds == v)


def cv_residuals(fold_params, X, resp, kfolds, labels="bgnmrv"):
    """V73:258-319: mfit.<model>.full = c(test$resp - predict(model_v, test)) over v = 1..nfolds."""
    cols = {lab: [] for lab in labels}
    for v, params in enumerate(fold_params, start=1):
        rows = holdout_rows(kfolds, v, X.shape[0])
        for lab in labels:
            cols[lab].append(resp[rows] - predict(params[lab], X[rows]))
    return np.column_stack([np.concatenate(cols[lab]) for lab in labels])


def optx_objective_literal(k, R):
    """machisplin.optimx.internal exactly as written at V73:329-331 (and 369-371 for four columns):
    sum(((r1*k1)/(k1+..+k6) + (r2*k2)/(k1+..+k6) + ...)^2)."""
    tot = 0.0
    for kk in k:
        tot = tot + kk
    acc = None
    for j, kk in enumerate(k):
        term = (R[:, j] * kk) / tot
        acc = term if acc is None else acc + term
    return float(np.sum(acc ** 2))
