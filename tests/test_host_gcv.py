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


def _random_band(m, bw, seed):
    rng = np.random.default_rng(seed)
    ab = np.zeros((m, bw + 1))
    ab[:, 0] = 10.0 + rng.uniform(0, 1, m)
    for d in range(1, bw + 1):
        ab[:m - d, d] = rng.uniform(-0.5, 0.5, m - d)
    return ab, rng.standard_normal(m)


def _band_eval(ab, g, lam):
    m, w = ab.shape
    lam_o, gcv, edf = C.c_double(), C.c_double(), C.c_double()
    q = np.empty(m)
    rc = _lib.load().mhs_host_gcv_band(ab.ctypes.data, w - 1, g.ctypes.data, m, m + 3, m + 3, 0.0, float(lam), 0,
                                       C.byref(lam_o), C.byref(gcv), C.byref(edf), q.ctypes.data)
    assert rc == 0
    return lam_o.value, gcv.value, edf.value, q


def test_fixed_bandwidth_path_matches_the_general_one():
    """Bandwidth 8 takes the compile-time-sized LDL' evaluation; the same matrix stored with a ninth, all-zero
    sub-diagonal takes the general banded Cholesky.  Both against a dense solve."""
    m = 300
    ab8, g = _random_band(m, 8, 5)
    ab9 = np.ascontiguousarray(np.column_stack([ab8, np.zeros(m)]))
    A = np.zeros((m, m))
    for d in range(9):
        A += np.diag(ab8[:m - d, d], -d)
    A = A + np.tril(A, -1).T
    for lam in (1e-3, 0.7, 40.0):
        _, gcv8, edf8, q8 = _band_eval(ab8, g, lam)
        _, gcv9, edf9, q9 = _band_eval(ab9, g, lam)
        assert abs(gcv8 - gcv9) < 1e-12 * abs(gcv9) and abs(edf8 - edf9) < 1e-10 * abs(edf9)
        assert np.abs(q8 - q9).max() < 1e-12 * np.abs(q9).max()
        Ai = np.linalg.inv(A + lam * np.eye(m))
        assert np.abs(q8 - Ai @ g).max() < 1e-11 * np.abs(q8).max()
        assert abs(edf8 - (3.0 + m - lam * np.trace(Ai))) < 1e-9 * m


def test_concurrent_gcv_searches_share_and_release_the_worker_pool():
    """One search leases the process-wide pool, the others build their own; every pool must wind down (a worker
    that misses the stop flag hangs the join) and every search must find the same lambda."""
    import threading
    ab, g = _random_band(500, 8, 11)
    ref = _band_eval(ab, g, float("nan"))[0]
    out, errs = [], []

    def work():
        try:
            for _ in range(8):
                out.append(_band_eval(ab, g, float("nan"))[0])
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=work) for _ in range(5)]
    [t.start() for t in ths]
    [t.join(timeout=120) for t in ths]
    assert not any(t.is_alive() for t in ths), "a GCV search did not return"
    # the pools differ in size (the process-wide one vs private ones of 16 threads): lambda must not -- the search
    # takes the same brackets whatever the thread count
    assert not errs and len(out) == 40 and all(v == ref for v in out)
