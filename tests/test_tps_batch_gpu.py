"""GPU parity of the batched small fits (csrc/tps_batch.hip, mhs_tps_fit_many): every spline of a reference-tiled
Step 3 (V73:690-738) fitted by ONE kernel launch, against the oracle's QR + eigen + GCV restatement and against the
library's one-fit-at-a-time route (mhs_tps_fit) on the same stations."""
import numpy as np
import pytest

from conftest import synth_stations
from oracle import tps as otps

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(b).max()


SIZES = [10, 33, 64, 97, 131, 160, 189, 193, 224, 225, 246, 256]


@pytest.mark.parametrize("mode", ["fields", "converged"])
def test_batch_gcv_fits_match_oracle_and_single_fit_route(hip, mode):
    sets = [synth_stations(n, 900 + n) for n in SIZES]
    fits = hip.tps.fit_many([s[0] for s in sets], [s[1] for s in sets], gcv_mode=mode)
    assert len(fits) == len(SIZES)
    for (xy, y), got, n in zip(sets, fits, SIZES):
        assert got is not None and got.n == n
        want = otps.fit(xy, y, gcv_mode=mode)
        tol = 1e-8 if mode == "fields" else 1e-6
        assert abs(got.lambda_ - want["lambda"]) / want["lambda"] < tol, n
        assert abs(got.eff_df - want["eff_df"]) < 1e-5 * want["eff_df"], n
        assert abs(got.gcv - want["gcv"]) < 1e-9 * want["gcv"], n
        ref = otps.fit(xy, y, lam=got.lambda_)
        assert _rel(got.c, ref["c"]) < 1e-8, n
        assert _rel(got.d, ref["d"]) < 1e-8, n
        assert np.array_equal(got.knots, want["knots"])
        # the library's own one-at-a-time route (one-block tridiagonalisation + host search)
        one = hip.Tps(xy, y, gcv_mode=mode)
        assert abs(got.lambda_ - one.lambda_) / one.lambda_ < (1e-10 if mode == "fields" else 1e-6), n
        if mode == "fields":
            assert _rel(got.c, one.c) < 1e-9 and _rel(got.d, one.d) < 1e-9, n
        # the knot records written by the kernel evaluate like the handle built from host coefficients
        rng = np.random.default_rng(n)
        pts = np.column_stack([rng.uniform(-78, -76, 300), rng.uniform(-7, -5, 300)])
        assert _rel(got.predict(pts), otps.predict_points(ref, pts)) < 1e-9, n


def test_batch_fixed_lambda(hip):
    sets = [synth_stations(n, 400 + n) for n in (12, 100, 200, 250)]
    lam = 3e-3
    fits = hip.tps.fit_many([s[0] for s in sets], [s[1] for s in sets], lambda_=lam)
    for (xy, y), got in zip(sets, fits):
        want = otps.fit(xy, y, lam=lam)
        assert got.lambda_ == lam
        assert _rel(got.c, want["c"]) < 1e-8 and _rel(got.d, want["d"]) < 1e-8
        assert abs(got.eff_df - want["eff_df"]) < 1e-8 * want["eff_df"]


def test_batch_replicates_mixed_sizes_and_failures(hip):
    """Replicated locations collapse inside the batch; a fit too large for a workgroup (n > 256) and one too small
    (n < 8) take mhs_tps_fit's route in the same call; a collinear set fails alone (None), the others are returned."""
    xy, y = synth_stations(150, 7)
    xy2 = np.vstack([xy, xy[:9], xy[3:5]])
    y2 = np.concatenate([y, y[:9] + 0.3, y[3:5] - 0.1])
    big = synth_stations(400, 11)
    tiny = synth_stations(6, 12)
    x = np.linspace(0, 1, 30)
    bad = (np.column_stack([x, 2 * x + 1]), np.sin(x))
    fits = hip.tps.fit_many([xy2, big[0], tiny[0], bad[0]], [y2, big[1], tiny[1], bad[1]])
    assert fits[3] is None
    assert fits[0].n == 150 and fits[1].n == 400 and fits[2].n == 6
    for (a, b), got in zip([(xy2, y2), big, tiny], fits[:3]):
        want = otps.fit(a, b)
        assert abs(got.lambda_ - want["lambda"]) / want["lambda"] < 1e-8
        ref = otps.fit(a, b, lam=got.lambda_)
        assert _rel(got.c, ref["c"]) < 1e-8 and _rel(got.d, ref["d"]) < 1e-8


def test_batch_many_more_fits_than_compute_units(hip):
    """600 fits on 256 compute units: workgroups loop over their fits; every result equals the fit done alone."""
    sets = [synth_stations(40 + (k % 7), 5000 + k) for k in range(600)]
    fits = hip.tps.fit_many([s[0] for s in sets], [s[1] for s in sets])
    for k in (0, 1, 255, 256, 257, 511, 599):
        alone = hip.tps.fit_many([sets[k][0]], [sets[k][1]])[0]
        assert fits[k].lambda_ == alone.lambda_ and np.array_equal(fits[k].c, alone.c) and np.array_equal(fits[k].d, alone.d)
    want = otps.fit(*sets[300])
    assert abs(fits[300].lambda_ - want["lambda"]) / want["lambda"] < 1e-8
