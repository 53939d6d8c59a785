"""Writes tests/golden/r_inputs/learn_fit.csv (200 rows: resp, a, b, LONG, LAT) and learn_fit_wts0.csv (the initial
nnet weights, nnet order) -- the inputs of the learner-fit section of capture_from_R.R and of the tests that read
its output.  Deterministic; run from the repository root:  python tests/golden/make_learn_inputs.py"""
import os

import numpy as np

rng = np.random.default_rng(20251017)
n = 200
a = rng.normal(10.0, 3.0, n)
b = rng.uniform(0.0, 40.0, n)
lon = rng.uniform(-78.0, -77.0, n)
lat = rng.uniform(-6.0, -5.0, n)
resp = 12.0 + 0.8 * a - 0.05 * b + 3.0 * np.sin(4.0 * (lon + 78.0)) + 2.0 * (lat + 5.5) ** 2 + rng.normal(0.0, 0.4, n)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "r_inputs")
np.savetxt(os.path.join(out, "learn_fit.csv"), np.column_stack([resp, a, b, lon, lat]), delimiter=",", fmt="%.17g",
           header="sigma=0.25\nresp,a,b,LONG,LAT", comments="# ")
np.savetxt(os.path.join(out, "learn_fit_wts0.csv"), rng.uniform(-0.7, 0.7, (4 + 1) * 10 + 11), fmt="%.17g")
