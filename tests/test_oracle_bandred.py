"""CPU: oracle/bandred.py -- the numpy restatement of the round-4 GCV route's pieces -- against dense linear algebra, so that
the GPU tests may lean on it."""
import numpy as np

from oracle import bandred as br
from oracle import tps as ot


def _tps_matrix(n, seed):
    rng = np.random.default_rng(seed)
    c = rng.choice(300 * 300, n, replace=False)
    u = np.column_stack([(c % 300 + 0.5) / 300, (c // 300 + 0.5) / 300])
    u = (u - u.min(0)) / (u.max(0) - u.min(0))
    y = np.sin(6 * u[:, 0]) * np.cos(5 * u[:, 1]) + 0.1 * rng.standard_normal(n)
    K = ot.gram(u)
    Q, _ = np.linalg.qr(np.column_stack([np.ones(n), u]), mode="complete")
    Q2 = Q[:, 3:]
    B = Q2.T @ K @ Q2
    return 0.5 * (B + B.T), Q2.T @ y


def test_householder_reconstruction_from_an_orthonormal_factor():
    rng = np.random.default_rng(0)
    P = rng.standard_normal((300, 32)) @ np.diag(10.0 ** rng.uniform(-3, 0, 32))
    V, T, R = br.cholqr2_householder(P)
    H = np.eye(300) - V @ T @ V.T
    assert np.max(np.abs(H.T @ H - np.eye(300))) < 1e-13
    HP = H.T @ P
    assert np.max(np.abs(HP[32:])) < 1e-13 * np.max(np.abs(P)) and np.max(np.abs(HP[:32] - R)) < 1e-13 * np.max(np.abs(P))
    assert np.allclose(np.tril(R, -1), 0) and np.allclose(np.triu(V[:32], 1), 0) and np.allclose(np.diag(V[:32]), 1)
    assert np.min(np.abs(1.0 / np.diag(np.linalg.inv(T)))) > 0          # T is invertible, upper triangular
    assert np.allclose(np.tril(T, -1), 0)


def test_expansion_and_basis_kernel_form():
    """The form the kernels use: chol(I + E) by its expansion, Yamamoto's V = [I; 0] - Q D, T = (I - (Q D)_top)^-T."""
    rng = np.random.default_rng(1)
    for t in (64, 300):
        P = rng.standard_normal((t, 32)) @ np.diag(10.0 ** rng.uniform(-3, 0, 32)) @ (np.eye(32) + 0.3 * rng.standard_normal((32, 32)))
        V, T, R = br.cholqr_expansion_yamamoto(P)
        H = np.eye(t) - V @ T @ V.T
        assert np.max(np.abs(H.T @ H - np.eye(t))) < 5e-14
        HP = H.T @ P
        assert np.max(np.abs(HP[32:])) < 1e-13 * np.max(np.abs(P)) and np.max(np.abs(HP[:32] - R)) < 1e-13 * np.max(np.abs(P))
        assert np.allclose(np.tril(R, -1), 0) and np.linalg.cond(V[:32]) < 20


def test_band_reduction_and_gcv_terms():
    B, g = _tps_matrix(430, 1)        # m = 427: ends with a short panel (t = 11)
    m = B.shape[0]
    ab, gq, panels, worst = br.band_reduce(B, g)
    assert worst < 1e6
    e, U = np.linalg.eigh(B)
    assert np.max(np.abs(np.linalg.eigvalsh(br.band_dense(ab)) - e)) < 1e-13 * e[-1]
    lam = 3e-5
    c2 = br.back_transform(br.band_solve(ab, gq, lam), panels)
    want = np.linalg.solve(B + lam * np.eye(m), g)
    assert np.max(np.abs(c2 - want)) < 1e-10 * np.max(np.abs(want))
    z = U.T @ g
    for lam in (1e-8, 1e-5, 1e-2, 10.0):
        neg, tr, q2 = br.band_gcv_terms(ab, gq, lam)
        assert neg == 0
        assert abs(tr - np.sum(1 / (e + lam))) < 1e-10 * tr
        assert abs(q2 - np.sum((z / (e + lam)) ** 2)) < 1e-9 * q2
    neg, _, _ = br.band_gcv_terms(ab, gq, -0.5 * (e[100] + e[101]))
    assert neg == 101
