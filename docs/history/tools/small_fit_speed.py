"""Phases of a small GCV fit (the reference-tiled mode's 130-250 stations per tile):  MHS_TIMING=1 python tools/small_fit_speed.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import machisplin_amd as m
m.init()
for n in (131, 200, 250):
    rng = np.random.default_rng(n)
    xy = rng.uniform(0, 1, (n, 2)); y = np.sin(6 * xy[:, 0]) * np.cos(5 * xy[:, 1]) + 0.1 * rng.standard_normal(n)
    m.Tps(xy, y)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); t = m.Tps(xy, y); best = min(best, time.perf_counter() - t0)
    print(f"n={n}: {best*1e3:.2f} ms  lambda={t.lambda_:.6g} c[0]={t.c[0]:.15g}", flush=True)
