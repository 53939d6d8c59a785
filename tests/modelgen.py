"""Test helpers: export scikit-learn estimators into the flat layouts of the reference's
fitted R objects (gbm / randomForest / ksvm / nnet), so the oracle can be cross-checked
against an independent evaluator of the same structure."""
import numpy as np


def sklearn_tree_to_gbm_nodes(tree, scale=1.0):
    """One sklearn regression tree -> gbm node arrays.  sklearn sends x <= thr left; gbm sends
    x < split left: identical whenever no x equals a threshold (thresholds are midpoints).
    gbm gives every split a MissingNode; it is appended as a terminal carrying the split
    node's own mean."""
    t = tree.tree_
    n = t.node_count
    var, val, left, right, missing = [], [], [], [], []
    extra = []
    for k in range(n):
        if t.children_left[k] == -1:
            var.append(-1); val.append(float(t.value[k, 0, 0]) * scale); left.append(0); right.append(0); missing.append(0)
        else:
            var.append(int(t.feature[k])); val.append(float(t.threshold[k]))
            left.append(int(t.children_left[k])); right.append(int(t.children_right[k]))
            missing.append(n + len(extra)); extra.append(float(t.value[k, 0, 0]) * scale)
    for v in extra:
        var.append(-1); val.append(v); left.append(0); right.append(0); missing.append(0)
    return var, val, left, right, missing


def gbm_from_sklearn(gbr, p):
    off, V, X, L, R, M = [0], [], [], [], [], []
    for est in gbr.estimators_[:, 0]:
        v, x, l, r, m = sklearn_tree_to_gbm_nodes(est, scale=gbr.learning_rate)
        V += v; X += x; L += l; R += r; M += m
        off.append(len(V))
    init = float(gbr.init_.constant_[0, 0])
    return {"kind": "gbm", "init_f": init, "tree_offsets": np.array(off, dtype=np.int64),
            "split_var": np.array(V, dtype=np.int32), "split_val": np.array(X), "left": np.array(L, dtype=np.int32),
            "right": np.array(R, dtype=np.int32), "missing": np.array(M, dtype=np.int32), "p": p}


def rf_from_sklearn(rf, p):
    off, L, R, S, V, SP, NP = [0], [], [], [], [], [], []
    for est in rf.estimators_:
        t = est.tree_
        for k in range(t.node_count):
            leaf = t.children_left[k] == -1
            L.append(0 if leaf else int(t.children_left[k]) + 1)
            R.append(0 if leaf else int(t.children_right[k]) + 1)
            S.append(-1 if leaf else -3)
            V.append(0 if leaf else int(t.feature[k]) + 1)
            SP.append(0.0 if leaf else float(t.threshold[k]))
            NP.append(float(t.value[k, 0, 0]))
        off.append(len(L))
    return {"kind": "rf", "tree_offsets": np.array(off, dtype=np.int64), "left": np.array(L, dtype=np.int32),
            "right": np.array(R, dtype=np.int32), "status": np.array(S, dtype=np.int32),
            "best_var": np.array(V, dtype=np.int32), "split": np.array(SP), "node_pred": np.array(NP), "p": p}


def svr_from_sklearn(svr, x_center, x_scale, y_center, y_scale):
    """sklearn SVR (rbf) fitted on standardised x, y -> kernlab layout.  sklearn:
    f = sum dual_coef_i K(sv_i, x) + intercept ; kernlab: f = sum alpha_i K - b."""
    return {"kind": "svr", "alpha": svr.dual_coef_[0].copy(), "sv": svr.support_vectors_.copy(),
            "b": -float(svr.intercept_[0]), "sigma": float(svr._gamma), "x_center": np.asarray(x_center),
            "x_scale": np.asarray(x_scale), "y_center": float(y_center), "y_scale": float(y_scale)}


def nnet_from_sklearn(mlp, y_scale, y_shift):
    """sklearn MLPRegressor(hidden_layer_sizes=(H,), activation='logistic') -> nnet wts order."""
    W1, W2 = mlp.coefs_
    b1, b2 = mlp.intercepts_
    p, H = W1.shape
    wts = []
    for h in range(H):
        wts += [b1[h]] + list(W1[:, h])
    wts += [b2[0]] + list(W2[:, 0])
    return {"kind": "nnet", "wts": np.array(wts), "p": p, "size": H, "y_scale": float(y_scale), "y_shift": float(y_shift)}
