"""TEST INFRASTRUCTURE ONLY (never imported by machisplin_amd/ or by bench.py's timed region).

numpy restatement of the ROUND-4 route of the GCV fit (machisplin_amd/csrc/tps_band32.hip), used by the tests to check the
HIP kernels piece by piece.  What it restates is this repo's own algorithm, not a reference file: fields::Tps (V73:722, V73:751)
only fixes WHAT is computed -- lambda by GCV over B = Q2'KQ2, then c and d -- and oracle/tps.py restates that with a dense
eigendecomposition.  The pieces here are the ones the GPU route is made of:

  cholqr2_householder   panel QR P = Q R by CholeskyQR2, then the compact-WY form H = I - V T V' of an orthogonal matrix whose
                        first b columns are Q D (D = diag(+-1)): LU of [I; 0] - Q D with the signs chosen so that every pivot
                        is >= 1 (Ballard, Demmel, Grigori, Jacquelin, Knight, Nguyen: "Reconstructing Householder vectors
                        from TSQR", 2015)
  band_reduce           blocked two-sided reduction of a symmetric matrix to bandwidth b with such panels
  band_gcv_terms        for one lambda, from the band alone and in ONE forward sweep: the inertia of T + lambda I, tr (T + lambda
                        I)^-1 and g'(T + lambda I)^-2 g, through an LDL' recurrence that carries its own derivative with respect
                        to lambda (tr M^-1 = d/dlambda log det M = sum d'_j / d_j, g'M^-2 g = -d/dlambda g'M^-1 g)
  band_solve            (T + lambda I) q = g
"""
from __future__ import annotations

import numpy as np


# ----------------------------------------------------------------------------------------------------- panel QR --
def cholqr2(P):
    """P = Q R with Q'Q = I to rounding (two Cholesky-QR passes).  Raises LinAlgError when P'P is not numerically SPD."""
    R1 = np.linalg.cholesky(P.T @ P).T
    Q1 = np.linalg.solve(R1.T, P.T).T
    R2 = np.linalg.cholesky(Q1.T @ Q1).T
    Q = np.linalg.solve(R2.T, Q1.T).T
    return Q, R2 @ R1


def householder_from_q(Q):
    """Q (t x b, orthonormal columns, t >= b) -> V (t x b unit lower trapezoidal), T (b x b upper triangular), D (b signs) with
    (I - V T V') [I_b; 0] = Q diag(D).  LU without pivoting of A = [I; 0] - Q diag(D), the sign D_k chosen when column k is
    reached (the elimination is linear in the column and leaves e_k alone): pivot = 1 + |q~_kk| >= 1."""
    t, b = Q.shape
    L = np.zeros((t, b))
    U = np.zeros((b, b))
    D = np.ones(b)
    for k in range(b):
        x = Q[:, k].copy()
        for j in range(k):
            x[j + 1:] -= L[j + 1:, j] * x[j]
        D[k] = -1.0 if x[k] >= 0 else 1.0
        U[:k, k] = -D[k] * x[:k]
        U[k, k] = 1.0 - D[k] * x[k]
        L[k, k] = 1.0
        L[k + 1:, k] = -D[k] * x[k + 1:] / U[k, k]
    T = np.linalg.solve(L[:b, :b], U.T).T          # T = U L1^-T
    return L, T, D


def cholqr2_householder(P):
    """Returns V, T, Rt with (I - V T V')' P = [Rt; 0], Rt upper triangular (its diagonal may carry either sign).  The LU
    form of the reconstruction (the first version of the GPU kernel; V unit lower trapezoidal, T upper triangular)."""
    Q, R = cholqr2(P)
    V, T, D = householder_from_q(Q)
    return V, T, D[:, None] * R


def _phi(X):
    return np.triu(X, 1) + 0.5 * np.diag(np.diag(X))


def cholqr_expansion_yamamoto(P):
    """The form the GPU kernels use now (b32_cholqr1/2_kernel + b32_tfin): first pass R1 = chol(P'P), Q1 = P R1^-1; the
    second pass through the EXPANSION of chol(I + E), E = Q1'Q1 - I: U = Phi(E) - Phi(Phi(E)'Phi(E)), R2 = I + U, R2^-1 =
    I - U + U^2 (both to O(E^3)); then Yamamoto's basis-kernel representation H = I - V T V', V = [I; 0] - Q D, T = (I -
    (Q D)_top)^-T with D_k = -sign(Q_kk): no triangular structure, no serial step.  Returns V, T, Rt as above."""
    t, b = P.shape
    R1 = np.linalg.cholesky(P.T @ P).T
    Q1 = np.linalg.solve(R1.T, P.T).T
    E = Q1.T @ Q1 - np.eye(b)
    E = 0.5 * (E + E.T)
    if not np.max(np.abs(E)) <= 1e-5:
        raise np.linalg.LinAlgError("panel too ill-conditioned for the expansion")
    U1 = _phi(E)
    U = U1 - _phi(U1.T @ U1)
    Q = Q1 @ (np.eye(b) - U + U @ U)
    D = np.where(np.diag(Q[:b]) >= 0, -1.0, 1.0)
    V = -(Q * D)
    V[:b] += np.eye(b)
    T = np.linalg.inv(V[:b]).T
    return V, T, D[:, None] * ((np.eye(b) + U) @ R1)


def householder_panel(P):
    """Classical Householder QR of a t x b panel with min(b, t - 1) reflectors (LAPACK dgeqr2 + dlarft), zero-padded to b
    columns: the form the short last panel takes, where t < b rules the Cholesky route out."""
    t, b = P.shape
    A = P.copy()
    V = np.zeros((t, b))
    tau = np.zeros(b)
    for j in range(min(b, t - 1)):
        x = A[j:, j]
        ss = float(x[1:] @ x[1:])
        alpha = x[0]
        if ss == 0.0:
            V[j, j] = 1.0
            continue
        beta = -np.copysign(np.sqrt(alpha * alpha + ss), alpha)
        tau[j] = (beta - alpha) / beta
        v = x / (alpha - beta)
        v[0] = 1.0
        V[j:, j] = v
        A[j:, j:] -= tau[j] * np.outer(v, v @ A[j:, j:])
    T = np.zeros((b, b))
    G = V.T @ V
    for j in range(b):
        T[j, j] = tau[j]
        if j:
            T[:j, j] = -tau[j] * (T[:j, :j] @ G[:j, j])
    R = np.triu(A[:b]) if t >= b else np.vstack([np.triu(A), np.zeros((b - t, b))])
    return V, T, R


# ------------------------------------------------------------------------------------------------ band reduction --
def band_reduce(B, g, b=32, small=None, form="yamamoto"):
    """B = Q Bb Q' with Bb of bandwidth b; returns the band ab[d, j] = Bb[j + d, j] (d = 0 .. b), g rotated to Q'g, the
    panels' (c, V, T) for the back-transform and the largest cond(P) met.  Panels with fewer than `small` (default b + 1) rows
    take the classical Householder form."""
    A = np.array(B, dtype=np.float64)
    g = np.array(g, dtype=np.float64)
    m = A.shape[0]
    small = b + 1 if small is None else small
    panels, worst = [], 1.0
    c = 0
    while m - c - b >= 2:
        r0, t = c + b, m - c - b
        P = A[r0:, c:c + b]
        if t >= small:
            sv = np.linalg.svd(P, compute_uv=False)
            worst = max(worst, sv[0] / sv[-1])
            V, T, R = cholqr_expansion_yamamoto(P) if form == "yamamoto" else cholqr2_householder(P)
        else:
            V, T, R = householder_panel(P)
        A[r0:, c:c + b] = 0.0
        A[r0:r0 + min(b, t), c:c + b] = R[:min(b, t)]
        A[c:c + b, r0:] = A[r0:, c:c + b].T
        A22 = A[r0:, r0:]
        Y = A22 @ V
        M = V.T @ Y
        W = Y @ T - 0.5 * V @ (T.T @ M @ T)
        A22 -= V @ W.T + W @ V.T
        g[r0:] -= V @ (T.T @ (V.T @ g[r0:]))
        panels.append((c, V, T))
        c += b
    ab = np.zeros((b + 1, m))
    for d in range(b + 1):
        ab[d, :m - d] = np.diagonal(A, -d)
    return ab, g, panels, worst


def back_transform(q, panels, b=32):
    """Q q for the Q of band_reduce."""
    r = np.array(q, dtype=np.float64)
    for c, V, T in reversed(panels):
        r0 = c + b
        r[r0:] -= V @ (T @ (V.T @ r[r0:]))
    return r


def band_dense(ab):
    b, m = ab.shape[0] - 1, ab.shape[1]
    A = np.zeros((m, m))
    for d in range(b + 1):
        i = np.arange(m - d)
        A[i + d, i] = ab[d, :m - d]
        A[i, i + d] = ab[d, :m - d]
    return A


# ------------------------------------------------------------------------------------ GCV terms from the band --
def band_gcv_terms(ab, g, lam):
    """One forward sweep over M = T + lam I (band ab, bandwidth b): LDL' column by column, right-looking, on a (b + 1)-wide
    window, carrying d/dlam of every quantity (M' = I).  Returns (negative pivots, tr M^-1, g'M^-2 g).  With y = L^-1 g:
    g'M^-1 g = sum y_j^2 / d_j, so g'M^-2 g = -sum (2 y_j y'_j d_j - y_j^2 d'_j) / d_j^2."""
    b, m = ab.shape[0] - 1, ab.shape[1]
    W = np.zeros((b + 1, m + b + 1))            # W[d, j]: column j of the part of M not yet eliminated
    W[:, :m] = ab
    W[0, :m] += lam
    dW = np.zeros_like(W)
    dW[0, :m] = 1.0
    y = np.concatenate([np.asarray(g, dtype=np.float64), np.zeros(b + 1)])
    dy = np.zeros_like(y)
    neg, tr, q2 = 0, 0.0, 0.0
    for j in range(m):
        d, dd = W[0, j], dW[0, j]
        if d == 0.0:
            d = -1e-300
        if d < 0:
            neg += 1
        a, da = W[1:, j].copy(), dW[1:, j].copy()       # column below the pivot
        l = a / d
        dl = (da - l * dd) / d
        tr += dd / d
        q2 -= (2.0 * y[j] * dy[j] * d - y[j] * y[j] * dd) / (d * d)
        # y[j + k] -= l_k y_j
        dy[j + 1:j + 1 + b] -= dl * y[j] + l * dy[j]
        y[j + 1:j + 1 + b] -= l * y[j]
        # column j + k, rows j + k .. j + b:  -= a_i l_k
        for k in range(1, b + 1):
            n = b + 1 - k
            dW[:n, j + k] -= da[k - 1:] * l[k - 1] + a[k - 1:] * dl[k - 1]
            W[:n, j + k] -= a[k - 1:] * l[k - 1]
    return neg, tr, q2


def band_solve(ab, g, lam):
    import scipy.linalg as sla
    a = np.array(ab)
    a[0] += lam
    return sla.solveh_banded(a, g, lower=True)
