"""Host side of SURVEY.md 8f rank 3: the ensemble weight search (V73:326-393) on CV residual columns."""
import numpy as np
import pytest

from oracle import ensemble as oe


def _residuals(seed, n=700, m=6):
    rng = np.random.default_rng(seed)
    common = rng.standard_normal(n)
    R = np.column_stack([0.6 * common * rng.uniform(0.2, 1.0) + rng.standard_normal(n) * rng.uniform(0.3, 2.0)
                         for _ in range(m)])
    return R


@pytest.mark.parametrize("m", [6, 4])
def test_objective_through_the_gram_matrix_is_the_reference_formula(m):
    from machisplin_amd import cv
    R = _residuals(1, m=m)
    G = R.T @ R
    rng = np.random.default_rng(2)
    for _ in range(20):
        k = rng.uniform(0.01, 1.0, m)
        a, b = cv.optx_objective(k, G), oe.optx_objective_literal(k, R)
        assert abs(a - b) <= 1e-12 * b
        assert abs(cv.optx_objective(3.7 * k, G) - a) <= 1e-12 * a      # scale invariance (V73:329-331)


@pytest.mark.parametrize("seed,smooth", [(3, False), (4, False), (5, True)])
def test_weight_search_reaches_a_constrained_minimum_and_applies_the_keep_rule(seed, smooth):
    from machisplin_amd import cv
    m = 4 if smooth else 6
    R = _residuals(seed, m=m)
    p, kept, wts, tot = cv.optx_weights(R, smooth_only=smooth)
    assert p.shape == (m,) and (p >= 0).all() and (p <= 1).all() and abs(tot - p.sum()) < 1e-15
    G = R.T @ R
    f = cv.optx_objective(p, G)
    assert f <= cv.optx_objective(np.full(m, 0.5), G) + 1e-12
    # no feasible point does better than 1e-6 relative: dense random search on the box plus the vertices' rays
    rng = np.random.default_rng(seed + 100)
    trial = np.vstack([rng.uniform(0, 1, (20000, m)), np.eye(m), p + 1e-3 * rng.standard_normal((2000, m))])
    trial = np.clip(trial, 0.0, 1.0)
    trial = trial[trial.sum(axis=1) > 1e-6]
    ft = np.einsum("ij,jk,ik->i", trial, G, trial) / trial.sum(axis=1) ** 2
    assert f <= ft.min() * (1 + 1e-6)
    labels = cv.ORDER_SMOOTH if smooth else cv.ORDER_ALL
    want = oe.select_weights(p, labels)
    assert (kept, wts, tot) == want
    for lab, w in zip(kept, wts):
        assert w == round(float(p[labels.index(lab)]), 2) and w > 0.05 * tot


def test_a_perfect_member_takes_all_the_weight():
    from machisplin_amd import cv
    R = _residuals(7)
    R[:, 4] = 0.0        # the forest predicts the hold-out rows exactly
    p, kept, wts, tot = cv.optx_weights(R)
    assert cv.optx_objective(p, R.T @ R) < 1e-10 * cv.optx_objective(np.full(6, 0.5), R.T @ R)
    assert "r" in kept


def test_holdout_rule_swaps_above_4000_rows():
    from machisplin_amd import cv
    k = np.tile(np.arange(1, 11), 500)
    assert np.array_equal(cv.holdout_rows(k[:4000], 3, 4000), np.flatnonzero(k[:4000] == 3))     # test on fold v
    assert np.array_equal(cv.holdout_rows(k, 3, 5000), np.flatnonzero(k != 3))                   # V73:228-230
    assert np.array_equal(cv.holdout_rows(k, 3, 5000), oe.holdout_rows(k, 3, 5000))


def test_oracle_lm_fit_known_answer():
    """oracle lm_fit (V73:252 / V73:600): exact recovery on noise-free data, intercept first."""
    from oracle import ensemble as oe
    rng = np.random.default_rng(4)
    X = rng.normal(size=(40, 3))
    beta = np.array([2.0, -1.0, 0.5, 3.0])
    coef = oe.lm_fit(X, beta[0] + X @ beta[1:])
    assert np.abs(coef - beta).max() < 1e-12
