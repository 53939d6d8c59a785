"""GPU: the pieces of the round-4 GCV route (machisplin_amd/csrc/tps_band32.hip, fields::Tps at V73:722 / V73:751) through
their C-ABI test hooks, against the numpy restatement in oracle/bandred.py and against dense linear algebra:
the 32-column-panel band reduction (CholeskyQR2 panels + Householder reconstruction, MFMA symmetric product and rank-64
update, short last panel), the twisted self-differentiating LDL' sweep that yields inertia / tr M^-1 / g'M^-2 g per lambda,
the banded solve and the back-transform."""
import ctypes as C

import numpy as np
import pytest

from oracle import bandred as br
from oracle import tps as ot

pytestmark = pytest.mark.gpu


def _tps_matrix(n, seed, cells=400):
    rng = np.random.default_rng(seed)
    c = rng.choice(cells * cells, n, replace=False)
    u = np.column_stack([(c % cells + 0.5) / cells, (c // cells + 0.5) / cells])
    u = (u - u.min(0)) / (u.max(0) - u.min(0))
    y = np.sin(6 * u[:, 0]) * np.cos(5 * u[:, 1]) + 0.1 * rng.standard_normal(n)
    K = ot.gram(u)
    Q, _ = np.linalg.qr(np.column_stack([np.ones(n), u]), mode="complete")
    Q2 = Q[:, 3:]
    B = Q2.T @ K @ Q2
    return 0.5 * (B + B.T), Q2.T @ y


def _reduce(hip, B, g, r=None):
    from machisplin_amd import _lib
    m = B.shape[0]
    Bf = np.asfortranarray(B)
    ab, gq, Qr = np.empty(m * 33), np.empty(m), np.empty(m)
    rr = np.ascontiguousarray(r if r is not None else np.zeros(m))
    bd = C.c_int(0)
    _lib.check(_lib.lib().mhs_band32_reduce(Bf.ctypes.data, g.ctypes.data, m, ab.ctypes.data, gq.ctypes.data, rr.ctypes.data,
                                            Qr.ctypes.data, C.byref(bd)))
    return ab.reshape(m, 33).T.copy(), gq, Qr, bd.value


def _terms(hip, ab, g, lam, deriv=True):
    from machisplin_amd import _lib
    m = ab.shape[1]
    abc = np.ascontiguousarray(ab.T)
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    neg, tr, q2 = np.empty(lam.size), np.empty(lam.size), np.empty(lam.size)
    _lib.check(_lib.lib().mhs_band32_gcv_terms(abc.ctypes.data, g.ctypes.data, m, lam.ctypes.data, lam.size, int(deriv),
                                               neg.ctypes.data, tr.ctypes.data, q2.ctypes.data))
    return neg, tr, q2


@pytest.mark.parametrize("m", [66, 97, 333, 1200])
def test_sweep_terms_match_the_eigenvalue_formulas_and_the_numpy_sweep(hip, m):
    """A random SPD band of width 32 (and g): inertia counts at shifts inside the spectrum against eigvalsh; tr M^-1 and
    g'M^-2 g at lambdas over ten decades against the spectral sums (what oracle/tps.py's GCV evaluates) and against the
    restated one-directional sweep."""
    rng = np.random.default_rng(m)
    ab = rng.standard_normal((33, m)) * np.exp(-0.15 * np.arange(33))[:, None]
    ab[0] = np.abs(ab[0]) + 3.0 * np.abs(ab[1:]).sum(0) / 2 + 0.5
    for d in range(1, 33):
        ab[d, m - d:] = 0.0
    g = rng.standard_normal(m)
    T = br.band_dense(ab)
    e, U = np.linalg.eigh(T)
    assert e[0] > 0
    z = U.T @ g
    lam = np.concatenate([10.0 ** np.arange(-6, 4.5, 0.75), [0.0]])
    neg, tr, q2 = _terms(hip, ab, g, lam)
    assert np.all(neg == 0)
    tr_ref = np.array([np.sum(1.0 / (e + l)) for l in lam])
    q2_ref = np.array([np.sum((z / (e + l)) ** 2) for l in lam])
    assert np.max(np.abs(tr - tr_ref) / tr_ref) < 1e-11
    assert np.max(np.abs(q2 - q2_ref) / q2_ref) < 1e-11
    for k in (0, 3, len(lam) - 2):
        _, tr1, q21 = br.band_gcv_terms(ab, g, lam[k])
        assert abs(tr[k] - tr1) < 1e-12 * tr1 and abs(q2[k] - q21) < 1e-12 * q21
    # inertia: shifts between eigenvalues (never closer than 1e-7 of the gap to one)
    ks = np.unique(np.linspace(0, m - 2, 9).astype(int))
    x = 0.5 * (e[ks] + e[ks + 1])
    cnt, _, _ = _terms(hip, ab, g, -x, deriv=False)
    assert np.array_equal(cnt, ks + 1.0)
    cnt2, _, _ = _terms(hip, ab, g, np.array([-(e[-1] * 1.0001), -(e[0] * 0.9999)]), deriv=False)
    assert np.array_equal(cnt2, [m, 0])


@pytest.mark.parametrize("m", [70, 401, 2050])
def test_banded_solve(hip, m):
    from machisplin_amd import _lib
    rng = np.random.default_rng(m + 1)
    ab = rng.standard_normal((33, m)) * 0.3
    ab[0] = np.abs(ab[1:]).sum(0) * 2 + 1.0
    for d in range(1, 33):
        ab[d, m - d:] = 0.0
    g = rng.standard_normal(m)
    abc = np.ascontiguousarray(ab.T)
    q = np.empty(m)
    _lib.check(_lib.lib().mhs_band32_solve(abc.ctypes.data, g.ctypes.data, m, 0.37, q.ctypes.data))
    want = br.band_solve(ab, g, 0.37)
    assert np.max(np.abs(q - want)) < 1e-12 * np.max(np.abs(want))


@pytest.mark.parametrize("n", [403, 900, 1003 + 32, 2600])
def test_band_reduction_keeps_the_spectrum_and_solves_the_system(hip, n):
    """B = Q2'KQ2 of n TPS stations on distinct cells: the reduced band has B's eigenvalues, Q'g its norm, Q Q'g = g, and
    c2 = Q (Bb + lambda I)^-1 Q'g solves (B + lambda I) c2 = g.  n - 3 mod 32 covers a short last panel (t < 64, the
    classical Householder kernel) in 1035 -> m = 1032 (t = 8) and 403 -> m = 400 (t = 16)."""
    B, g = _tps_matrix(n, n)
    m = B.shape[0]
    ab, gq, _, bd = _reduce(hip, B, g)
    assert bd == 0
    e = np.linalg.eigvalsh(B)
    e2 = np.linalg.eigvalsh(br.band_dense(ab))
    assert np.max(np.abs(e2 - e)) < 2e-13 * e[-1]
    assert abs(np.linalg.norm(gq) - np.linalg.norm(g)) < 1e-12 * np.linalg.norm(g)
    _, _, back, _ = _reduce(hip, B, g, r=gq)
    assert np.max(np.abs(back - g)) < 5e-12 * np.max(np.abs(g))
    lam = 1e-4
    q = br.band_solve(ab, gq, lam)
    _, _, c2, _ = _reduce(hip, B, g, r=q)
    want = np.linalg.solve(B + lam * np.eye(m), g)
    assert np.max(np.abs(c2 - want)) < 1e-9 * np.max(np.abs(want))


def test_rank_deficient_panel_is_reported(hip):
    """Two identical stations' columns make the first panel rank deficient: the Cholesky-QR must say so (the fit then takes
    the 8-column Householder route) instead of returning garbage."""
    B, g = _tps_matrix(500, 3)
    B[:, 5] = B[:, 4]
    B[5, :] = B[4, :]
    _, _, _, bd = _reduce(hip, B, g)
    assert bd == 1
