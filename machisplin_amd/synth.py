"""Deterministic synthetic inputs of the BASELINE.json shapes:
grid geometry of the bundled rasters (1/1200 degree cells, NW origin -78, -5), stations on
distinct cell centres, and the TPS-only residual  r = sin(6u) cos(5v) + 0.1 N(0,1).
Used by bench.py, __graft_entry__.smoke() and the tests; no reference data involved."""
from __future__ import annotations

import numpy as np

from .raster import Geometry

BASE_SEED = 20251017


def grid(nrow: int, ncol: int) -> Geometry:
    return Geometry(-78.0, -5.0, 1.0 / 1200.0, 1.0 / 1200.0, nrow, ncol)


def stations(geom: Geometry, n: int, seed: int):
    """n stations on distinct cell centres (knots are cell centres, V73:128-133,145);
    returns xy (n x 2: LONG, LAT), the cell rows/cols, and unit-square coordinates."""
    rng = np.random.default_rng(seed)
    if geom.ncell <= 50_000_000:
        cells = rng.choice(geom.ncell, size=n, replace=False)
    else:  # a permutation of a 4e8-cell grid would need gigabytes: oversample, keep the first n distinct
        draw = rng.integers(0, geom.ncell, size=int(n * 1.2) + 64)
        _, first = np.unique(draw, return_index=True)
        cells = draw[np.sort(first)][:n]
        assert cells.size == n
    rows, cols = np.divmod(cells, geom.ncol)
    xy = np.column_stack([geom.x_from_col(cols), geom.y_from_row(rows)])
    uv = np.column_stack([(cols + 0.5) / geom.ncol, (rows + 0.5) / geom.nrow])
    return xy, rows, cols, uv


def tps_residual(uv: np.ndarray, seed: int) -> np.ndarray:
    rng = np.random.default_rng(seed + 1)
    return np.sin(6 * uv[:, 0]) * np.cos(5 * uv[:, 1]) + 0.1 * rng.standard_normal(uv.shape[0])


# ---------------------------------------------------------------- covariate planes --
COV_RANGES = [(76.0, 4668.0), (-1.0, 877.0), (-207.0, 152.0), (0.0, 360.0), (-50.0, 50.0),
              (0.0, 1.0), (10.0, 3000.0), (-5.0, 40.0)]  # alt, slope, TWI (extdata aux.xml), then generic


def _cov_layer(rng, col, row, k, torch):
    """layer k's field at the (broadcastable) unit coordinates col, row; consumes the layer's random draws"""
    a = rng.uniform(0.3, 1.0, 6)
    f = rng.uniform(0.5, 6.0, 6)
    gq = rng.uniform(0.5, 6.0, 6)
    ph = rng.uniform(0, 2 * np.pi, 6)
    z = None
    for m in range(6):
        term = a[m] * torch.sin(2 * np.pi * (f[m] * col + gq[m] * row) + ph[m])
        z = term if z is None else z + term
    lo, hi = COV_RANGES[k % len(COV_RANGES)]
    return (z / a.sum() * 0.5 + 0.5) * (hi - lo) + lo


def covariates(geom: Geometry, n_layers: int, seed: int, dtype: str = "f32", nodata_frac: float = 0.0, window=None):
    """cov_k = sum_{m<6} a_km sin(2 pi (f_km col/ncol + g_km row/nrow) + phi_km), rescaled to
    alt/slope/TWI-like ranges.  Built on the GPU; returns a (C, nrow, ncol)
    device tensor (float32, float64, or int16 with NoData -32768 on `nodata_frac` of cells)
    and the NoData value.  `window` = (r0, r1, c0, c1) builds only that crop of the grid's planes (the field is a
    function of the absolute cell indices, so a crop equals the slice of the full planes)."""
    import torch
    from . import _lib
    dev = torch.device("cuda", _lib.init())
    rng = np.random.default_rng(seed + 7)
    r0, r1, c0, c1 = window if window is not None else (0, geom.nrow, 0, geom.ncol)
    if window is not None and nodata_frac > 0:
        raise ValueError("nodata_frac is only supported for whole-grid planes")
    col = (torch.arange(c0, c1, device=dev, dtype=torch.float64) / geom.ncol)[None, :]
    row = (torch.arange(r0, r1, device=dev, dtype=torch.float64) / geom.nrow)[:, None]
    tdt = {"f32": torch.float32, "f64": torch.float64, "i16": torch.int16}[dtype]
    out = torch.empty((n_layers, r1 - r0, c1 - c0), dtype=tdt, device=dev)
    for k in range(n_layers):
        z = _cov_layer(rng, col, row, k, torch)
        if dtype == "i16":
            z = torch.round(z)
        out[k] = z.to(tdt)
        del z
    nodata = float("nan")
    if dtype == "i16":
        nodata = -32768.0
    if nodata_frac > 0:
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed + 11)
        mask = torch.rand((geom.nrow, geom.ncol), device=dev, generator=gen) < nodata_frac
        for k in range(n_layers):
            out[k][mask] = -32768 if dtype == "i16" else float("nan")
    return out, nodata


def covariates_at(geom: Geometry, n_layers: int, seed: int, rows, cols, dtype: str = "f32") -> np.ndarray:
    """The same planes sampled at the given cells (n x C float64, after the plane dtype's rounding): what
    terra::extract(rast_stack, xy) returns at the stations, without building the planes."""
    import torch
    from . import _lib
    dev = torch.device("cuda", _lib.init())
    rng = np.random.default_rng(seed + 7)
    col = torch.from_numpy(np.asarray(cols, dtype=np.float64)).to(dev) / geom.ncol
    row = torch.from_numpy(np.asarray(rows, dtype=np.float64)).to(dev) / geom.nrow
    tdt = {"f32": torch.float32, "f64": torch.float64, "i16": torch.int16}[dtype]
    out = []
    for k in range(n_layers):
        z = _cov_layer(rng, col, row, k, torch)
        if dtype == "i16":
            z = torch.round(z)
        out.append(z.to(tdt).to(torch.float64).cpu().numpy())
    return np.column_stack(out)


# ---------------------------------------------- trainer-free ensemble parameter sets --
# Structurally faithful stand-ins for the fitted CRAN objects (same array layouts, sizes and
# tree shapes as a real fit at the BASELINE sizes), generated from the seed in seconds so the
# GPU box needs neither R nor a long scikit-learn fit.  The STRUCTURES are drawn at random (which leaf is split on
# which predictor at which station's value, which stations are support vectors); the VALUES are fitted cheaply to
# the response -- stagewise residual means per leaf for the boosted trees, kernel ridge at the support vectors for
# the SVR, a few hundred L-BFGS steps for the network, least squares for the linear and MARS members -- so that the
# ensemble explains the response as a real one does (R^2 ~ 0.9; README.md:56 quotes > 0.99 for real fits) and the
# spline downstream sees a realistic residual.

def response(X: np.ndarray, uv: np.ndarray, seed: int) -> np.ndarray:
    """y = 250 - 0.0055 alt + 3 sin(4u) + 2 cos(3v) + N(0,1)   (SURVEY.md 8d)."""
    rng = np.random.default_rng(seed + 3)
    return 250.0 - 0.0055 * X[:, 0] + 3 * np.sin(4 * uv[:, 0]) + 2 * np.cos(3 * uv[:, 1]) + rng.standard_normal(X.shape[0])


def lm_params(X, y):
    A = np.column_stack([np.ones(X.shape[0]), X])
    return {"kind": "lm", "coef": np.linalg.lstsq(A, y, rcond=None)[0]}


def nnet_params(X, y, seed, size=10):
    rng = np.random.default_rng(seed + 21)
    p = X.shape[1]
    mu, sd = X.mean(0), X.std(0) + 1e-12
    wts = []
    for _ in range(size):  # unscaled inputs (V73:463): weights ~ 1/sd so units are not all saturated
        w = rng.standard_normal(p) / sd
        wts += [float(-(w * mu).sum() + rng.standard_normal())] + list(w)
    wts += list(rng.standard_normal(size + 1) * 0.3)
    ymin = float(y.min())
    y_scale = float((y - ymin).max())
    wts = _nnet_train(np.array(wts), X, (y - ymin) / (y_scale if y_scale > 0 else 1.0), size, mu, sd)
    return {"kind": "nnet", "wts": wts, "p": p, "size": size, "y_scale": y_scale, "y_shift": ymin}


def _nnet_train(w0, X, t, size, mu, sd, iters=300, max_rows=4000):
    """A few hundred L-BFGS steps on nnet's least-squares criterion (linear output, V73:463), in standardised
    coordinates for conditioning; the weights are mapped back to the raw inputs nnet's predict sees."""
    from scipy.optimize import minimize
    n, p = X.shape
    if n > max_rows:      # the structure, not the last digit of the fit, is what matters: cap the cost
        sel = np.random.default_rng(12345).choice(n, max_rows, replace=False)
        X, t = X[sel], t[sel]
    Z = (X - mu) / sd
    W1 = w0[:size * (p + 1)].reshape(size, p + 1).copy()
    W1[:, 0] += W1[:, 1:] @ mu                 # standardised inputs: b' = b + w.mu, w' = w * sd
    W1[:, 1:] *= sd
    theta0 = np.concatenate([W1.ravel(), w0[size * (p + 1):]])

    def loss(theta):
        A = theta[:size * (p + 1)].reshape(size, p + 1)
        v = theta[size * (p + 1):]
        H = 1.0 / (1.0 + np.exp(-np.clip(A[:, 0][None, :] + Z @ A[:, 1:].T, -30, 30)))
        r = v[0] + H @ v[1:] - t
        gH = np.outer(r, v[1:]) * H * (1.0 - H)
        gA = np.column_stack([gH.sum(0), gH.T @ Z])
        return 0.5 * float(r @ r), np.concatenate([gA.ravel(), [r.sum()], H.T @ r])

    res = minimize(loss, theta0, jac=True, method="L-BFGS-B", options={"maxiter": iters})
    A = res.x[:size * (p + 1)].reshape(size, p + 1).copy()
    A[:, 1:] /= sd
    A[:, 0] -= A[:, 1:] @ mu
    return np.concatenate([A.ravel(), res.x[size * (p + 1):]])


def earth_params(X, y, seed, nterms=15):
    rng = np.random.default_rng(seed + 22)
    n, p = X.shape
    dirs = np.zeros((nterms, p), dtype=np.int32)
    cuts = np.zeros((nterms, p))
    B = [np.ones(n)]
    for k in range(1, nterms):
        v = int(rng.integers(p))
        d = int(rng.choice([1, -1, 2], p=[0.45, 0.45, 0.1]))
        c = float(X[rng.integers(n), v]) if d != 2 else 0.0
        dirs[k, v], cuts[k, v] = d, c
        B.append(X[:, v] if d == 2 else np.maximum(0.0, d * (X[:, v] - c)))
    coef = np.linalg.lstsq(np.column_stack(B), y, rcond=None)[0]
    return {"kind": "earth", "coef": coef, "dirs": dirs, "cuts": cuts}


def svr_params(X, y, seed, frac_sv=0.6):
    rng = np.random.default_rng(seed + 23)
    n, p = X.shape
    mu, sd = X.mean(0), X.std(0, ddof=1)
    nsv = max(4, int(frac_sv * n))
    idx = rng.choice(n, nsv, replace=False)
    sv = (X[idx] - mu) / sd
    sigma = float(rng.uniform(0.15, 0.4))
    # kernel ridge at the support vectors: (K_ss + mu I) alpha = y~_s, b = 0 -- the dual coefficients of a
    # least-squares SVM on the same kernel; ksvm's differ in value (eps-insensitive loss, |alpha| <= C), not in
    # count or layout
    ys = (y[idx] - y.mean()) / y.std(ddof=1)
    sq = (sv * sv).sum(1)
    K = np.exp(-sigma * np.maximum(sq[:, None] + sq[None, :] - 2.0 * (sv @ sv.T), 0.0))
    K[np.diag_indices(nsv)] += 0.1
    alpha = np.linalg.solve(K, ys)
    del K
    return {"kind": "svr", "alpha": alpha, "sv": sv, "b": 0.0,
            "sigma": sigma, "x_center": mu, "x_scale": sd,
            "y_center": float(y.mean()), "y_scale": float(y.std(ddof=1))}


def gbm_params(X, y, seed, n_trees=10000, n_splits=5, shrinkage=0.001):
    """interaction.depth = 5 trees (V73:493): 5 splits grown on random leaves, every split
    owning a left, right and missing child (gbm's node layout: 1 + 3*5 = 16 nodes)."""
    rng = np.random.default_rng(seed + 24)
    n, p = X.shape
    npt = 1 + 3 * n_splits
    sd_y = float(y.std())
    split_var = np.full((n_trees, npt), -1, dtype=np.int32)
    split_val = rng.standard_normal((n_trees, npt)) * sd_y * shrinkage
    left = np.zeros((n_trees, npt), dtype=np.int32)
    right = np.zeros((n_trees, npt), dtype=np.int32)
    missing = np.zeros((n_trees, npt), dtype=np.int32)
    # leaves[t, :] candidate terminal nodes that may still be split (left/right children only)
    nleaf = np.ones(n_trees, dtype=np.int64)
    leaves = np.zeros((n_trees, 1 + 2 * n_splits), dtype=np.int64)
    T = np.arange(n_trees)
    for s in range(n_splits):
        pick = (rng.random(n_trees) * nleaf).astype(np.int64)
        node = leaves[T, pick]
        v = rng.integers(0, p, n_trees)
        thr = X[rng.integers(0, n, n_trees), v]
        split_var[T, node] = v
        split_val[T, node] = thr
        l, r, m = 1 + 3 * s, 2 + 3 * s, 3 + 3 * s
        left[T, node], right[T, node], missing[T, node] = l, r, m
        leaves[T, pick] = l          # the split leaf is replaced by its left child ...
        leaves[T, nleaf] = r         # ... and the right child is appended
        nleaf += 1
    _gbm_boost_leaves(X, y, split_var, split_val, left, right, n_splits)
    off = np.arange(n_trees + 1, dtype=np.int64) * npt
    return {"kind": "gbm", "init_f": float(y.mean()), "tree_offsets": off, "split_var": split_var.ravel(),
            "split_val": split_val.ravel(), "left": left.ravel(), "right": right.ravel(),
            "missing": missing.ravel(), "p": p}


def _gbm_boost_leaves(X, y, split_var, split_val, left, right, n_splits, max_rows=5000):
    """Terminal values by stagewise boosting on the given (random) structures: tree t's leaf value is the shrunken
    mean of the current residual over the training rows that reach the leaf (gbm's gaussian terminal-node estimate),
    and the residual is updated before tree t+1.  The shrinkage is scaled so that the whole sequence removes most of
    what such trees can explain whatever its length (0.001 x 10 000 trees in the reference's final fit, V73:493)."""
    n_trees, npt = split_var.shape
    n = X.shape[0]
    if n > max_rows:
        sel = np.random.default_rng(4321).choice(n, max_rows, replace=False)
        X, y = X[sel], y[sel]
        n = max_rows
    shrink = min(0.5, 30.0 / n_trees)
    resid = y - y.mean()
    rows = np.arange(n)
    for t0 in range(0, n_trees, 512):              # leaf of every training row in a block of trees, level by level
        t1 = min(n_trees, t0 + 512)
        node = np.zeros((t1 - t0, n), dtype=np.int64)
        T = np.arange(t0, t1)[:, None]
        for _ in range(n_splits):
            v = split_var[T, node]
            inner = v >= 0
            xv = X[rows[None, :], np.where(inner, v, 0)]
            nxt = np.where(xv < split_val[T, node], left[T, node], right[T, node])
            node = np.where(inner, nxt, node)
        for t in range(t0, t1):
            leaf = node[t - t0]
            cnt = np.bincount(leaf, minlength=npt)
            val = shrink * np.bincount(leaf, weights=resid, minlength=npt) / np.maximum(cnt, 1)
            term = split_var[t] < 0
            split_val[t, term] = np.where(cnt[term] > 0, val[term], 0.0)   # unreached terminals (missing children): 0
            resid = resid - val[leaf]


def rf_params(X, y, seed, n_trees=500, nodesize=5):
    """randomForest-shaped regression trees: bootstrap sample, recursive axis splits at the
    midpoint of two sample values until nodes hold <= nodesize points; node numbering in
    creation order as randomForest does (leftDaughter = ncur+1, rightDaughter = ncur+2)."""
    rng = np.random.default_rng(seed + 25)
    n, p = X.shape
    offs, L, R, S, V, SP, NP = [0], [], [], [], [], [], []
    for _ in range(n_trees):
        boot = rng.integers(0, n, n)
        Xb, yb = X[boot], y[boot]
        left, right, status, var, split, pred = [0], [0], [-1], [0], [0.0], [float(yb.mean())]
        stack = [(0, np.arange(n))]
        while stack:
            k, idx = stack.pop()
            if idx.size <= nodesize:
                continue
            for _try in range(4):
                v = int(rng.integers(p))
                xv = Xb[idx, v]
                a, b = xv[rng.integers(idx.size)], xv[rng.integers(idx.size)]
                if a != b:
                    break
            else:
                continue
            thr = 0.5 * (a + b)
            go_left = xv <= thr
            il, ir = idx[go_left], idx[~go_left]
            if il.size == 0 or ir.size == 0:
                continue
            nl = len(left)
            for sub in (il, ir):
                left.append(0); right.append(0); status.append(-1); var.append(0); split.append(0.0)
                pred.append(float(yb[sub].mean()))
            left[k], right[k], status[k], var[k], split[k] = nl + 1, nl + 2, -3, v + 1, float(thr)
            stack.append((nl, il))
            stack.append((nl + 1, ir))
        offs.append(offs[-1] + len(left))
        L += left; R += right; S += status; V += var; SP += split; NP += pred
    return {"kind": "rf", "tree_offsets": np.array(offs, dtype=np.int64), "left": np.array(L, dtype=np.int32),
            "right": np.array(R, dtype=np.int32), "status": np.array(S, dtype=np.int32),
            "best_var": np.array(V, dtype=np.int32), "split": np.array(SP), "node_pred": np.array(NP), "p": p}


def ensemble_params(X, y, seed, n_gbm_trees=10000, n_rf_trees=500, which="bgnmrv"):
    """Parameter dicts in the reference's model order b, g, n, m, r, v (V73:340-362)."""
    makers = {"b": lambda: gbm_params(X, y, seed, n_trees=n_gbm_trees), "g": lambda: lm_params(X, y),
              "n": lambda: nnet_params(X, y, seed), "m": lambda: earth_params(X, y, seed),
              "r": lambda: rf_params(X, y, seed, n_trees=n_rf_trees), "v": lambda: svr_params(X, y, seed)}
    return [makers[k]() for k in which]


OPTX_WEIGHTS = (0.31, 0.22, 0.12, 0.18, 0.27, 0.41)  # p1..p6 of the L-BFGS-B fit


def mean_tree_visits(prm: dict, X: np.ndarray) -> float:
    """Mean number of split-node visits per sample over all trees of a gbm / rf parameter set
    (host-side walk over a small sample; feeds bench.py's algorithmic node-visit count)."""
    X = np.asarray(X, dtype=np.float64)
    n = X.shape[0]
    off = prm["tree_offsets"]
    rows = np.arange(n)
    visits = 0
    gbm = prm["kind"] == "gbm"
    for t in range(off.size - 1):
        o = off[t]
        node = np.zeros(n, dtype=np.int64)
        while True:
            idx = o + node
            active = (prm["split_var"][idx] >= 0) if gbm else (prm["status"][idx] != -1)
            if not active.any():
                break
            ia = idx[active]
            visits += int(active.sum())
            if gbm:
                x = X[rows[active], prm["split_var"][ia]]
                node[active] = np.where(np.isnan(x), prm["missing"][ia],
                                        np.where(x < prm["split_val"][ia], prm["left"][ia], prm["right"][ia]))
            else:
                x = X[rows[active], prm["best_var"][ia] - 1]
                node[active] = np.where(x <= prm["split"][ia], prm["left"][ia], prm["right"][ia]) - 1
    return visits / float(n)
