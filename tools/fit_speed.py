import time, numpy as np, sys
sys.path.insert(0, '.'); import os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import machisplin_amd as m
m.init()
for n in (500, 2000, 5000):
    rng = np.random.default_rng(n)
    xy = rng.uniform(0, 1, (n, 2)); y = np.sin(6*xy[:,0])*np.cos(5*xy[:,1]) + 0.1*rng.standard_normal(n)
    for lam in (1e-3, None):
        t0 = time.time(); t = m.Tps(xy, y, lambda_=lam); dt = time.time() - t0
        t0 = time.time(); t = m.Tps(xy, y, lambda_=lam); dt = time.time() - t0
        print(f"n={n} lambda={lam}: {dt*1e3:.1f} ms  lam={t.lambda_:.4g} chol GFLOP/s={(n-3)**3/3/dt/1e9:.1f}")
