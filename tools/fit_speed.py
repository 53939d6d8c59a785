"""Wall time of mhs_tps_fit by route and size (run on the GPU box):  python tools/fit_speed.py [sizes...]
fixed lambda = Gram + projection + MFMA blocked Cholesky + solves; GCV = band reduction + host search + back-transform.
Rates on the model counts of SURVEY.md 8d: (n-3)^3/3 for the Cholesky route, 4/3 (n-3)^3 for the GCV route."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import machisplin_amd as m  # noqa: E402

m.init()
sizes = [int(a) for a in sys.argv[1:]] or [500, 2000, 5000, 10000, 20000]
for n in sizes:
    rng = np.random.default_rng(n)
    xy = rng.uniform(0, 1, (n, 2))
    y = np.sin(6 * xy[:, 0]) * np.cos(5 * xy[:, 1]) + 0.1 * rng.standard_normal(n)
    for lam in (1e-3, None):
        best = 1e30
        for _ in range(3):
            t0 = time.perf_counter()
            t = m.Tps(xy, y, lambda_=lam)
            best = min(best, time.perf_counter() - t0)
        mm = n - 3
        flop = mm ** 3 / 3 if lam is not None else 4 * mm ** 3 / 3
        print(f"n={n:6d} {'fixed' if lam is not None else 'GCV  '}: {best * 1e3:9.2f} ms   lambda={t.lambda_:.4g}   "
              f"{flop / best / 1e12:7.3f} TF/s on the {'(n-3)^3/3' if lam is not None else '4/3 (n-3)^3'} model "
              f"({100 * flop / best / 78.6e12:.1f} % of the 78.6 TF FP64 peak)", flush=True)
    if n <= 3000:   # the Cholesky route against a dense host solve of the same saddle-point system
        u = (xy - xy.min(0)) / (xy.max(0) - xy.min(0))
        d2 = ((u[:, None, :] - u[None, :, :]) ** 2).sum(-1)
        K = np.where(d2 > 0, 0.5 / (8 * np.pi) * d2 * np.log(np.maximum(d2, 1e-300)), 0.0)
        T = np.column_stack([np.ones(n), u])
        M = np.block([[K + 1e-3 * np.eye(n), T], [T.T, np.zeros((3, 3))]])
        sol = np.linalg.solve(M, np.concatenate([y, np.zeros(3)]))
        t = m.Tps(xy, y, lambda_=1e-3)
        print(f"         fixed-lambda coefficients vs dense host solve: rel err c {np.abs(t.c - sol[:n]).max() / np.abs(sol[:n]).max():.2e}"
              f"  d {np.abs(t.d - sol[n:]).max() / np.abs(sol[n:]).max():.2e}", flush=True)
