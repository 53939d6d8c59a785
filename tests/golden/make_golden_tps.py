"""Regenerates tests/golden/tps_*.npz.  Run from the repo root IN THE BUILD CONTAINER:

    python tests/golden/make_golden_tps.py

Inputs: (a) the reference's bundled station table /root/reference/data-raw/sampling.csv
(813 rows: long, lat, bio_1, bio_12 -- data, not code; config 1 of BASELINE.json) and
(b) seeded synthetic stations.  Expected outputs come from the
oracle (oracle/tps.py) and, independently, from scipy.interpolate.RBFInterpolator
(thin_plate_spline, degree=1, smoothing = 8 pi lambda on range-scaled coordinates).
The reference itself (R + fields) cannot run here, so these vectors pin the oracle against
an independent implementation, not against R: PARITY UNPINNED.
"""
import os

import numpy as np
from scipy.interpolate import RBFInterpolator

import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import tps  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def grid_points(xmin, ymax, res, nrow, ncol):
    x, y = tps.cell_centres(xmin, ymax, res, res, nrow, ncol)
    X, Y = np.meshgrid(x, y)
    return np.column_stack([X.ravel(), Y.ravel()])


def case(name, xy, y, lam, geom):
    m = tps.fit(xy, y, lam=lam)
    mg = tps.fit(xy, y)              # gcv, fields mode
    mc = tps.fit(xy, y, gcv_mode="converged")
    xmin, ymax, res, nrow, ncol = geom
    surf = tps.predict_grid(m, xmin, ymax, res, res, nrow, ncol)
    rb = RBFInterpolator(m["knots"], m["yM"], kernel="thin_plate_spline", degree=1,
                         smoothing=8 * np.pi * lam)
    pts = (grid_points(xmin, ymax, res, nrow, ncol) - m["center"]) / m["scale"]
    surf_scipy = rb(pts).reshape(nrow, ncol)
    np.savez_compressed(
        os.path.join(HERE, f"tps_{name}.npz"), xy=xy, y=y, lam=lam, geom=np.array(geom, dtype=np.float64),
        c=m["c"], d=m["d"], center=m["center"], scale=m["scale"], eff_df=m["eff_df"], gcv=m["gcv"],
        surf=surf, surf_scipy=surf_scipy,
        lam_gcv_fields=mg["lambda"], gcv_fields=mg["gcv"], eff_df_fields=mg["eff_df"],
        lam_gcv_converged=mc["lambda"], c_gcv=mg["c"], d_gcv=mg["d"])
    print(name, "n", xy.shape[0], "lam_gcv", mg["lambda"], "scipy rel",
          np.abs(surf - surf_scipy).max() / np.abs(surf).max())


def main():
    rng = np.random.default_rng(20251017)
    for n in (12, 200):
        xy = np.column_stack([rng.uniform(-78, -76, n), rng.uniform(-7, -5, n)])
        u = (xy - xy.min(0)) / (xy.max(0) - xy.min(0))
        y = np.sin(6 * u[:, 0]) * np.cos(5 * u[:, 1]) + 0.1 * rng.standard_normal(n)
        case(f"synth{n}", xy, y, 2e-3, (-78.0, -5.0, 2.0 / 64, 48, 64))
    ref_csv = "/root/reference/data-raw/sampling.csv"
    tab = np.loadtxt(ref_csv, delimiter=",", skiprows=1)
    xy = tab[:, :2]
    y = tab[:, 2] - tab[:, 2].mean()  # bio_1 centred as a pseudo-residual
    case("sampling813", xy, y, 1e-3, (xy[:, 0].min() - 0.05, xy[:, 1].max() + 0.05, 0.05, 48, 64))


if __name__ == "__main__":
    main()
