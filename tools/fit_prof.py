"""GCV (or fixed-lambda) fits of n stations under a profiler: python tools/fit_prof.py [n] [gcv|fixed] [repeats]"""
import numpy as np, sys
import os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import machisplin_amd as m
m.init()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
fixed = len(sys.argv) > 2 and sys.argv[2] == "fixed"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rng = np.random.default_rng(n)
xy = rng.uniform(0, 1, (n, 2)); y = np.sin(6*xy[:,0])*np.cos(5*xy[:,1]) + 0.1*rng.standard_normal(n)
for _ in range(reps): t = m.Tps(xy, y, lambda_=1e-3 if fixed else None)
