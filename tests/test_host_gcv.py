"""CPU: the library's host-side GCV search on the tridiagonal form (no GPU needed)
against the oracle's eigenvalue form of fields' criterion."""
import ctypes as C

import numpy as np
import pytest
import scipy.linalg as sl

from machisplin_amd import _lib
from oracle import tps


def _tridiag_problem(n, seed, reps=0):
    rng = np.random.default_rng(seed)
    xy = rng.uniform(0, 1, (n, 2))
    y = np.sin(6 * xy[:, 0]) * np.cos(5 * xy[:, 1]) + 0.1 * rng.standard_normal(n)
    if reps:
        xy = np.vstack([xy, xy[:reps]])
        y = np.concatenate([y, y[:reps] + 0.2])
    return xy, y


@pytest.mark.parametrize("n,reps,mode", [(40, 0, "fields"), (300, 0, "fields"), (300, 0, "converged"), (120, 5, "fields")])
def test_tridiagonal_gcv_matches_eigen_gcv(n, reps, mode):
    xy, y = _tridiag_problem(n, n + reps, reps)
    m = tps.fit(xy, y, gcv_mode=mode)
    u, w = m["knots"], m["weightsM"]
    sw = np.sqrt(w)
    K = sw[:, None] * tps.gram(u) * sw[None, :]
    T = sw[:, None] * np.column_stack([np.ones(n), u])
    Q, _ = np.linalg.qr(T, mode="complete")
    Q2 = Q[:, 3:]
    B = Q2.T @ K @ Q2
    B = 0.5 * (B + B.T)
    H, P = sl.hessenberg(B, calc_q=True)
    a = np.ascontiguousarray(np.diag(H))
    b = np.ascontiguousarray(np.diag(H, -1))
    g = np.ascontiguousarray(P.T @ (Q2.T @ (sw * m["yM"])))
    lam, gcv, edf = C.c_double(), C.c_double(), C.c_double()
    q = np.empty(n - 3)
    lib = _lib.load()
    rc = lib.mhs_host_gcv_tridiag(a.ctypes.data, b.ctypes.data, g.ctypes.data, n - 3, n, m["N"],
                                  m["pure_ss"], float("nan"), {"fields": 0, "converged": 1}[mode],
                                  C.byref(lam), C.byref(gcv), C.byref(edf), q.ctypes.data)
    assert rc == 0
    tol = 1e-8 if mode == "fields" else 1e-6
    assert abs(lam.value - m["lambda"]) < tol * m["lambda"]
    assert abs(gcv.value - m["gcv"]) < 1e-9 * m["gcv"]
    assert abs(edf.value - m["eff_df"]) < 1e-5 * m["eff_df"]
    # q = (T + lam I)^-1 g  back-transformed is the oracle's c at the same lambda
    ref = tps.fit(xy, y, lam=lam.value)
    c = sw * (Q2 @ (P @ q))
    assert np.abs(c - ref["c"]).max() < 1e-8 * np.abs(ref["c"]).max()


def test_fixed_lambda_evaluation_only():
    xy, y = _tridiag_problem(60, 1)
    m = tps.fit(xy, y, lam=0.01)
    e, z = m["eig"], m["z"]
    a = np.ascontiguousarray(e[::-1].copy())  # a diagonal "tridiagonal" matrix
    b = np.zeros(len(e) - 1)
    g = np.ascontiguousarray(z[::-1].copy())
    lam, gcv, edf = C.c_double(), C.c_double(), C.c_double()
    rc = _lib.load().mhs_host_gcv_tridiag(a.ctypes.data, b.ctypes.data, g.ctypes.data, len(e), 60, 60, 0.0,
                                          0.01, 0, C.byref(lam), C.byref(gcv), C.byref(edf), None)
    assert rc == 0 and lam.value == 0.01
    assert abs(gcv.value - m["gcv"]) < 1e-12 * m["gcv"] and abs(edf.value - m["eff_df"]) < 1e-9


@pytest.mark.parametrize("n,bw,mode", [(60, 8, "fields"), (400, 8, "fields"), (400, 8, "converged"), (90, 3, "fields")])
def test_banded_gcv_matches_eigen_gcv(n, bw, mode):
    """The GPU fit reduces Q2'KQ2 only to a BAND; the host evaluates fields' criterion on it
    (banded Cholesky + Takahashi trace).  Here the band form is produced with numpy."""
    xy, y = _tridiag_problem(n, 3 * n)
    m = tps.fit(xy, y, gcv_mode=mode)
    u = m["knots"]
    Q, _ = np.linalg.qr(np.column_stack([np.ones(n), u]), mode="complete")
    Q2 = Q[:, 3:]
    B = Q2.T @ tps.gram(u) @ Q2
    A = 0.5 * (B + B.T)
    mm = n - 3
    P = np.eye(mm)
    for c in range(0, mm - bw - 1, bw):  # blocked Householder reduction to bandwidth bw
        q, _ = np.linalg.qr(A[c + bw:, c:c + bw], mode="complete")
        Qf = np.eye(mm)
        Qf[c + bw:, c + bw:] = q
        A = Qf.T @ A @ Qf
        P = P @ Qf
    assert np.abs(np.tril(A, -bw - 1)).max() < 1e-12 * np.abs(A).max()
    ab = np.zeros((mm, bw + 1))
    for d in range(bw + 1):
        ab[:mm - d, d] = np.diag(A, -d)
    g = np.ascontiguousarray(P.T @ (Q2.T @ y))
    lam, gcv, edf = C.c_double(), C.c_double(), C.c_double()
    q = np.empty(mm)
    rc = _lib.load().mhs_host_gcv_band(ab.ctypes.data, bw, g.ctypes.data, mm, n, n, 0.0, float("nan"),
                                       {"fields": 0, "converged": 1}[mode], C.byref(lam), C.byref(gcv), C.byref(edf),
                                       q.ctypes.data)
    assert rc == 0
    assert abs(lam.value - m["lambda"]) < (1e-8 if mode == "fields" else 1e-6) * m["lambda"]
    assert abs(gcv.value - m["gcv"]) < 1e-9 * m["gcv"] and abs(edf.value - m["eff_df"]) < 1e-5 * m["eff_df"]
    ref = tps.fit(xy, y, lam=lam.value)
    assert np.abs(Q2 @ (P @ q) - ref["c"]).max() < 1e-8 * np.abs(ref["c"]).max()
