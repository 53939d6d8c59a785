"""CPU: pin the TPS oracle against the committed golden vectors, against scipy's
independent RBF implementation, and against the analytic known-answer cases of
SURVEY.md section 8c (G1-G4).  (The reference has no tests of its own: parity unpinned.)"""
import glob
import os

import numpy as np
import pytest
from scipy.interpolate import RBFInterpolator

from oracle import tps

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "tps_*.npz"))))
def test_oracle_reproduces_golden(path):
    z = np.load(path)
    m = tps.fit(z["xy"], z["y"], lam=float(z["lam"]))
    assert np.allclose(m["c"], z["c"], rtol=0, atol=1e-9 * np.abs(z["c"]).max())
    assert np.allclose(m["d"], z["d"], rtol=0, atol=1e-9 * np.abs(z["d"]).max())
    xmin, ymax, res, nrow, ncol = z["geom"]
    surf = tps.predict_grid(m, xmin, ymax, res, res, int(nrow), int(ncol))
    scale = np.abs(z["surf"]).max()
    assert np.abs(surf - z["surf"]).max() < 1e-10 * scale
    assert np.abs(surf - z["surf_scipy"]).max() < 1e-10 * scale  # independent implementation
    mg = tps.fit(z["xy"], z["y"])
    assert abs(mg["lambda"] - float(z["lam_gcv_fields"])) < 1e-8 * float(z["lam_gcv_fields"])
    mc = tps.fit(z["xy"], z["y"], gcv_mode="converged")
    assert abs(mc["lambda"] - float(z["lam_gcv_converged"])) < 1e-6 * float(z["lam_gcv_converged"])
    assert mc["gcv"] <= mg["gcv"] * (1 + 1e-12)  # the converged search is at least as low


def test_eigen_route_equals_direct_saddle_point_and_scipy():
    rng = np.random.default_rng(3)
    xy = rng.uniform(0, 5, (300, 2))
    y = np.cos(xy[:, 0]) + 0.05 * rng.standard_normal(300)
    m = tps.fit(xy, y)
    md = tps.fit_direct(xy, y, m["lambda"])
    assert np.abs(m["c"] - md["c"]).max() < 1e-10 * np.abs(m["c"]).max()
    assert np.abs(m["d"] - md["d"]).max() < 1e-10 * np.abs(m["d"]).max()
    pts = rng.uniform(0, 5, (1000, 2))
    rb = RBFInterpolator(m["knots"], y, kernel="thin_plate_spline", degree=1, smoothing=8 * np.pi * m["lambda"])
    ref = rb((pts - m["center"]) / m["scale"])
    assert np.abs(tps.predict_points(m, pts) - ref).max() < 1e-11 * np.abs(ref).max()
    assert np.abs(np.column_stack([np.ones(300), m["knots"]]).T @ m["c"]).max() < 1e-9  # T'c = 0


def test_linear_data_gives_plane():
    rng = np.random.default_rng(4)
    xy = rng.uniform(-1, 1, (60, 2))
    y = 1.0 + 2.0 * xy[:, 0] - 3.0 * xy[:, 1]
    m = tps.fit(xy, y, lam=1e-2)
    assert np.abs(m["c"]).max() < 1e-10
    pts = rng.uniform(-1, 1, (50, 2))
    assert np.abs(tps.predict_points(m, pts) - (1 + 2 * pts[:, 0] - 3 * pts[:, 1])).max() < 1e-10


def test_lambda_limits():
    rng = np.random.default_rng(5)
    xy = rng.uniform(0, 1, (80, 2))
    y = np.sin(4 * xy[:, 0]) + xy[:, 1] ** 2
    m0 = tps.fit(xy, y, lam=1e-12)
    assert np.abs(tps.predict_points(m0, xy) - y).max() < 1e-6  # interpolation
    minf = tps.fit(xy, y, lam=1e12)
    A = np.column_stack([np.ones(80), xy])
    plane = A @ np.linalg.lstsq(A, y, rcond=None)[0]
    assert np.abs(tps.predict_points(minf, xy) - plane).max() < 1e-6  # least-squares plane


def test_replicates_collapse_to_weighted_problem():
    rng = np.random.default_rng(6)
    xy = rng.uniform(0, 1, (50, 2))
    y = rng.standard_normal(50)
    xy2 = np.vstack([xy, xy[:4]])
    y2 = np.concatenate([y, y[:4] + 1.0])
    m = tps.fit(xy2, y2, lam=1e-2)
    assert m["knots"].shape == (50, 2) and m["N"] == 54
    assert np.allclose(m["weightsM"][:4], 2.0) and np.allclose(m["weightsM"][4:], 1.0)
    assert np.isclose(m["pure_ss"], 4 * 2 * 0.25)
    md = tps.fit_direct(xy2, y2, 1e-2)
    assert np.abs(m["c"] - md["c"]).max() < 1e-10 * np.abs(m["c"]).max()


def test_degenerate_inputs():
    x = np.linspace(0, 1, 20)
    with pytest.raises(ValueError):
        tps.fit(np.column_stack([x, x]), x)
    with pytest.raises(ValueError):
        tps.fit(np.array([[0, 0], [1, 0], [0, 1.0]]), np.arange(3.0))


def test_phi_floor_and_grid_convention():
    assert tps.radial_phi(np.array([0.0]))[0] == tps.radial_phi(np.array([1e-20]))[0]
    x, y = tps.cell_centres(-78.0, -5.0, 0.5, 0.25, 4, 3)
    assert np.allclose(x, [-77.75, -77.25, -76.75]) and np.allclose(y, [-5.125, -5.375, -5.625, -5.875])
