"""Oracle: thin-plate smoothing spline fit + evaluation as `fields::Tps` does it.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED vs R.

Reference call sites (V73 = /root/reference/R/ensemble.machine.learning.thin.
plate.splines.V73.R):
  * fit:   ``fields::Tps(MyTPSdata[2:3], MyTPSdata[1])``   V73:722 (per tile)
           ``fields::Tps(dat_tps[[i]][,c(n.covars,n.covars+1)], res.FINAL)`` V73:751
  * eval:  ``terra::interpolate(terra::rast(rb), mod.tps.elev)``  V73:726, V73:753

`fields` is a CRAN dependency (DESCRIPTION:11, unpinned, not vendored).  The
algorithm restated here is the published one of fields' ``Tps -> Krig ->
Krig.engine.default -> gcv.Krig -> Krig.coef`` chain and ``predict.Krig``
(SURVEY.md section 8a rows a1/a2 and Appendix B):

  u      = (x - min_col(x)) / (max_col(x) - min_col(x))          scale.type="range"
  phi(d2)= (1/(8 pi)) * 0.5*log(max(d2,1e-20)) * max(d2,1e-20)    radbas.constant(2,2)*radfun
  K_ij   = phi(|u_i-u_j|^2) ; T = [1,u1,u2] ; T = [Q1 Q2] R
  B      = Q2' K Q2 = U diag(e) U' ; z = U' Q2' y
  GCV(l) = (RSS(l)/n + pure_ss/(N-n)) / (1 - trA(l)/n)^2
           RSS = sum((l z_i/(e_i+l))^2), trA = 3 + sum(e_i/(e_i+l))
  c      = Q2 U diag(1/(e+l)) z ; d = R^-1 Q1' (y - K c)
  f(x,y) = d0 + d1 u + d2 v + sum_j c_j phi(|(u,v)-u_j|^2)

Replicated locations are collapsed to weighted means first (Krig.replicates);
with weights w the system is (K + l W^-1) c + T d = yM, T'c = 0.
"""
from __future__ import annotations

import numpy as np

EIGHT_PI = 8.0 * np.pi
NT = 3  # dimension of the polynomial null space for d=2, m=2


def radial_phi(d2: np.ndarray) -> np.ndarray:
    """(1/8pi) r^2 log r with fields' 1e-20 floor on d2 (radfun in fields' radbas.f)."""
    d2 = np.maximum(d2, 1e-20)
    return (0.5 / EIGHT_PI) * np.log(d2) * d2


def collapse_replicates(xy: np.ndarray, y: np.ndarray):
    """fields' Krig.replicates: unique locations in first-appearance order,
    yM = mean of replicates, weightsM = replicate count (unit input weights),
    pure_ss = within-replicate sum of squares."""
    xy = np.asarray(xy, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    _, first, inv = np.unique(xy, axis=0, return_index=True, return_inverse=True)
    inv = np.asarray(inv).reshape(-1)
    order = np.argsort(first)  # unique groups by first appearance
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    gid = rank[inv]
    n = order.size
    cnt = np.bincount(gid, minlength=n).astype(np.float64)
    ym = np.bincount(gid, weights=y, minlength=n) / cnt
    pure_ss = float(np.sum((y - ym[gid]) ** 2))
    xm = xy[first[order]]
    return xm, ym, cnt, pure_ss


def range_scale(xm: np.ndarray):
    center = xm.min(axis=0)
    scale = xm.max(axis=0) - center
    return center, scale


def gram(u: np.ndarray) -> np.ndarray:
    dx = u[:, None, 0] - u[None, :, 0]
    dy = u[:, None, 1] - u[None, :, 1]
    return radial_phi(dx * dx + dy * dy)


def _gcv_terms(lam, e, z, n, N, pure_ss):
    lam = np.atleast_1d(np.asarray(lam, dtype=np.float64))[:, None]
    rss = np.sum((lam * z[None, :] / (e[None, :] + lam)) ** 2, axis=1)
    tra = NT + np.sum(e[None, :] / (e[None, :] + lam), axis=1)
    mse = rss / n
    if N - n > 0:
        mse = mse + pure_ss / (N - n)
    den = 1.0 - tra / n
    with np.errstate(divide="ignore", invalid="ignore"):
        g = np.where(den > 0, mse / den ** 2, np.nan)
    return g, tra


def gcv_value(lam: float, e, z, n, N, pure_ss) -> float:
    return float(_gcv_terms(lam, e, z, n, N, pure_ss)[0][0])


def lambda_grid(e: np.ndarray, n: int, nstep: int = 200) -> np.ndarray:
    """gcv.Krig's default lambda grid: log-spaced between the lambda whose
    effective df drops below nt+0.05 and the one whose df exceeds 0.95 n."""
    D = 1.0 / e  # increasing, since e is sorted decreasing
    l1 = 1.0 / D[0]
    for _ in range(20):
        tr = NT + np.sum(1.0 / (1.0 + l1 * D))
        if tr < NT + 0.05:
            break
        l1 *= 4.0
    l2 = 1.0 / D[-1]
    for _ in range(20):
        tr = NT + np.sum(1.0 / (1.0 + l2 * D))
        if tr > 0.95 * n:
            break
        l2 /= 4.0
    return np.exp(np.linspace(np.log(l2), np.log(l1), nstep))


def golden_section(ax, bx, cx, f, tol, niter=25):
    """fields' golden.section.search (Numerical-Recipes bracketing golden section
    that stops when |f2-f1| < tol)."""
    r = 0.61803399
    con = 1.0 - r
    x0, x3 = ax, cx
    if abs(cx - bx) > abs(bx - ax):
        x1, x2 = bx, bx + con * (cx - bx)
    else:
        x2, x1 = bx, bx - con * (bx - ax)
    f1, f2 = f(x1), f(x2)
    for _ in range(niter):
        if f2 < f1:
            x0, x1, x2 = x1, x2, r * x2 + con * x3
            f1, f2 = f2, f(x2)
        else:
            x3, x2, x1 = x2, x1, r * x1 + con * x0
            f2, f1 = f1, f(x1)
        if abs(f2 - f1) < tol:
            break
    return (x1, f1) if f1 < f2 else (x2, f2)


def find_lambda(e, z, n, N, pure_ss, mode="fields"):
    """mode "fields": 200-point grid + golden section with tol = 0.01*GCVmin
    (gcv.Krig default tol=.01, Krig.find.gcvmin).  mode "converged": same
    bracket, golden section on log(lambda) to 1e-13 relative."""
    grid = lambda_grid(e, n)
    g, _ = _gcv_terms(grid, e, z, n, N, pure_ss)
    ok = ~np.isnan(g)
    grid, g = grid[ok], g[ok]
    il = int(np.argmin(g))
    if il == 0 or il == grid.size - 1:
        return float(grid[il])
    f = lambda lam: gcv_value(lam, e, z, n, N, pure_ss)
    if mode == "fields":
        lam, _ = golden_section(grid[il - 1], grid[il], grid[il + 1], f, tol=0.01 * g[il])
        return float(lam)
    if mode == "converged":
        a, b = np.log(grid[il - 1]), np.log(grid[il + 1])
        r = 0.5 * (np.sqrt(5.0) - 1.0)
        x1, x2 = b - r * (b - a), a + r * (b - a)
        f1, f2 = f(np.exp(x1)), f(np.exp(x2))
        for _ in range(200):
            if f1 < f2:
                b, x2, f2 = x2, x1, f1
                x1 = b - r * (b - a)
                f1 = f(np.exp(x1))
            else:
                a, x1, f1 = x1, x2, f2
                x2 = a + r * (b - a)
                f2 = f(np.exp(x2))
            if abs(b - a) < 1e-13:
                break
        return float(np.exp(0.5 * (a + b)))
    raise ValueError(mode)


def fit(xy, y, lam=None, gcv_mode="fields"):
    """fields::Tps(x, Y) restated.  Returns a dict with the fields of the Krig
    object that predict.Krig consumes (c, d, lambda, knots, transform)."""
    xm, ym, w, pure_ss = collapse_replicates(xy, y)
    N = int(np.asarray(y).size)
    n = xm.shape[0]
    if n <= NT:
        raise ValueError("Tps needs more than 3 distinct locations")
    center, scale = range_scale(xm)
    if np.any(scale <= 0):
        raise ValueError("degenerate station coordinates (zero range)")
    u = (xm - center) / scale
    sw = np.sqrt(w)
    K = gram(u)
    T = np.column_stack([np.ones(n), u[:, 0], u[:, 1]])
    Kt = sw[:, None] * K * sw[None, :]
    Tt = sw[:, None] * T
    Q, R = np.linalg.qr(Tt, mode="complete")
    if abs(R[2, 2]) < 1e-10 * abs(R[0, 0]):
        raise ValueError("collinear station coordinates")
    Q1, Q2 = Q[:, :NT], Q[:, NT:]
    B = Q2.T @ Kt @ Q2
    B = 0.5 * (B + B.T)
    e, U = np.linalg.eigh(B)
    e, U = e[::-1], U[:, ::-1]  # decreasing, as R's eigen()
    yt = sw * ym
    z = U.T @ (Q2.T @ yt)
    if lam is None:
        lam = find_lambda(e, z, n, N, pure_ss, mode=gcv_mode)
    lam = float(lam)
    ct = Q2 @ (U @ (z / (e + lam)))
    d = np.linalg.solve(R[:NT, :NT], Q1.T @ (yt - Kt @ ct - lam * ct))
    c = sw * ct
    g, tra = _gcv_terms(lam, e, z, n, N, pure_ss)
    return {
        "c": c, "d": d, "lambda": lam, "center": center, "scale": scale,
        "knots": u, "xM": xm, "yM": ym, "weightsM": w, "eff_df": float(tra[0]),
        "gcv": float(g[0]), "eig": e, "z": z, "N": N, "pure_ss": pure_ss,
    }


def fit_direct(xy, y, lam):
    """Independent route for cross-checking `fit`: dense saddle-point solve of
    [[K + lam W^-1, T],[T', 0]] [c; d] = [yM; 0]."""
    xm, ym, w, _ = collapse_replicates(xy, y)
    n = xm.shape[0]
    center, scale = range_scale(xm)
    u = (xm - center) / scale
    K = gram(u)
    T = np.column_stack([np.ones(n), u[:, 0], u[:, 1]])
    A = np.zeros((n + NT, n + NT))
    A[:n, :n] = K + lam * np.diag(1.0 / w)
    A[:n, n:] = T
    A[n:, :n] = T.T
    sol = np.linalg.solve(A, np.concatenate([ym, np.zeros(NT)]))
    return {"c": sol[:n], "d": sol[n:], "lambda": float(lam), "center": center,
            "scale": scale, "knots": u}


def predict_points(model, xy, block=4096):
    """predict.Krig: fields.mkpoly(x,2) %*% d + Rad.cov(x, knots, C=c)."""
    xy = np.asarray(xy, dtype=np.float64).reshape(-1, 2)
    uv = (xy - model["center"]) / model["scale"]
    kn, c, d = model["knots"], model["c"], model["d"]
    out = np.empty(uv.shape[0])
    for s in range(0, uv.shape[0], block):
        p = uv[s:s + block]
        dx = p[:, None, 0] - kn[None, :, 0]
        dy = p[:, None, 1] - kn[None, :, 1]
        out[s:s + block] = d[0] + d[1] * p[:, 0] + d[2] * p[:, 1] + radial_phi(dx * dx + dy * dy) @ c
    return out


def cell_centres(xmin, ymax, xres, yres, nrow, ncol, r0=0, r1=None, c0=0, c1=None):
    """terra xFromCol / yFromRow: x = xmin + (col+0.5)*xres, y = ymax - (row+0.5)*yres,
    rows counted from the north edge (terra cell order, V73:128-133)."""
    r1 = nrow if r1 is None else r1
    c1 = ncol if c1 is None else c1
    x = xmin + (np.arange(c0, c1, dtype=np.float64) + 0.5) * xres
    y = ymax - (np.arange(r0, r1, dtype=np.float64) + 0.5) * yres
    return x, y


def predict_grid(model, xmin, ymax, xres, yres, nrow, ncol, r0=0, r1=None, c0=0, c1=None):
    """terra::interpolate(geometry-only raster, tps): every cell centre in the
    window [r0,r1) x [c0,c1) is evaluated; row-major from the north-west cell."""
    x, y = cell_centres(xmin, ymax, xres, yres, nrow, ncol, r0, r1, c0, c1)
    X, Y = np.meshgrid(x, y)
    out = predict_points(model, np.column_stack([X.ravel(), Y.ravel()]))
    return out.reshape(y.size, x.size)
