"""Writes tests/golden/r_inputs/members_crop.csv and members_stations.csv -- the inputs of the member-predict section of
capture_from_R.R (gbm / randomForest / ksvm fitted on stations and predicted on a raster crop, V73:497 / 521-523 / 582-584) and
of the tests that read its output.  The crop is 48 rows x 64 columns of the reference's bundled TWI / slope overviews
(tests/golden/cfg1_extdata.npz, rows 744.., columns 222..: no NoData there) plus a synthetic alt plane (alt.tif is not in the
repository), with LONG / LAT at the cell centres; the 300 stations are cells of a 400 x 400 neighbourhood of the crop.
Deterministic; run from the repository root:  python tests/golden/make_member_inputs.py"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
z = np.load(os.path.join(HERE, "cfg1_extdata.npz"))
xmin, ymax, xres, yres = (float(v) for v in z["geom"][:4])
R0, C0, NR, NC = 744, 222, 48, 64                                       # a 448 x 464 neighbourhood (544.., 22..) holds no NoData
twi, slope = z["TWI"].astype(np.float64), z["slope"].astype(np.float64)
assert not (twi[R0 - 200:R0 + 248, C0 - 200:C0 + 264] == float(z["nodata"])).any() and not (slope[R0 - 200:R0 + 248, C0 - 200:C0 + 264] == float(z["nodata"])).any()


def planes(rows, cols):
    u, v = cols / 1632.0, rows / 1238.0
    alt = np.round(2400.0 + 900.0 * np.sin(5.0 * u + 1.0) * np.cos(4.0 * v) + 300.0 * np.sin(17.0 * u * v))
    return np.column_stack([alt, slope[rows, cols], twi[rows, cols], xmin + (cols + 0.5) * xres, ymax - (rows + 0.5) * yres])


rr, cc = np.meshgrid(np.arange(R0, R0 + NR), np.arange(C0, C0 + NC), indexing="ij")
crop = planes(rr.ravel(), cc.ravel())                                   # terra cell order
rng = np.random.default_rng(20251018)
cells = rng.choice(400 * 400, 300, replace=False)
sr, sc = R0 - 176 + cells // 400, C0 - 168 + cells % 400
X = planes(sr, sc)
resp = np.round(250.0 - 0.0055 * X[:, 0] * 10 + 0.02 * X[:, 1] - 0.05 * X[:, 2] + 30.0 * np.sin(40.0 * (X[:, 3] - xmin))
                + rng.normal(0.0, 2.0, X.shape[0]))                  # integers, as the bundled bio_1
out = os.path.join(HERE, "r_inputs")
hdr = "nrow=%d ncol=%d sigma=0.2\n%s" % (NR, NC, "alt,slope,TWI,LONG,LAT")
np.savetxt(os.path.join(out, "members_crop.csv"), crop, delimiter=",", fmt="%.17g", header=hdr, comments="# ")
np.savetxt(os.path.join(out, "members_stations.csv"), np.column_stack([resp, X]), delimiter=",", fmt="%.17g",
           header="n=300\nresp,alt,slope,TWI,LONG,LAT", comments="# ")
print("wrote members_crop.csv", crop.shape, "members_stations.csv", X.shape)
