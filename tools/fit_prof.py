import numpy as np, sys
import os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import machisplin_amd as m
m.init()
n = 5000
rng = np.random.default_rng(n)
xy = rng.uniform(0, 1, (n, 2)); y = np.sin(6*xy[:,0])*np.cos(5*xy[:,1]) + 0.1*rng.standard_normal(n)
for _ in range(2): t = m.Tps(xy, y)
