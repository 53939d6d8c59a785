"""Torch-free driver for rocprofv3 (kernel trace and PMC passes) on the spline fit:  python tools/fit_pmc.py ROUTE N [REPS]
ROUTE = fixed (MFMA blocked Cholesky) | gcv (band reduction + host search) | gcvcache (one full GCV fit, then REPS refits of other
responses on the same stations through the reduction cache)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import machisplin_amd as m  # noqa: E402

route, n = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
m.init()
rng = np.random.default_rng(n)
xy = rng.uniform(0, 1, (n, 2))
y = np.sin(6 * xy[:, 0]) * np.cos(5 * xy[:, 1]) + 0.1 * rng.standard_normal(n)
if route == "gcvcache":      # round 3: other response layers on the same stations reuse the band reduction (mhs_tps_reduction_cache)
    from machisplin_amd import tps
    with tps.reduction_cache():
        t = m.Tps(xy, y)                                     # builds the entry
        for k in range(reps):
            t = m.Tps(xy, y + 0.1 * (k + 1) * xy[:, 0])      # Q'g through band_qt_kernel, host search, back-transform
else:
    for _ in range(reps):
        t = m.Tps(xy, y, lambda_=1e-3 if route == "fixed" else None)
print(route, n, "lambda", t.lambda_)
