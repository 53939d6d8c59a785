"""Batched small fits: timing of mhs_tps_fit_many against one-at-a-time mhs_tps_fit (49 station sets of the cfg3 tile sizes)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import machisplin_amd as m
m.init()
rng = np.random.default_rng(1)
sizes = list(rng.integers(131, 225, 49))
sets = []
for n in sizes:
    xy = rng.uniform(0, 1, (n, 2)); y = np.sin(6 * xy[:, 0]) * np.cos(5 * xy[:, 1]) + 0.1 * rng.standard_normal(n)
    sets.append((xy, y))
xs = [s[0] for s in sets]; ys = [s[1] for s in sets]
for rep in range(4):
    t0 = time.perf_counter(); fits = m.tps.fit_many(xs, ys); t1 = time.perf_counter()
    print(f"fit_many of {len(sets)} fits (n = {min(sizes)}..{max(sizes)}): {(t1 - t0) * 1e3:.2f} ms", flush=True)
t0 = time.perf_counter(); one = [m.Tps(x, y) for x, y in sets]; t1 = time.perf_counter()
print(f"one at a time: {(t1 - t0) * 1e3:.2f} ms")
err = max(abs(a.lambda_ - b.lambda_) / b.lambda_ for a, b in zip(fits, one))
errc = max(np.abs(a.c - b.c).max() / np.abs(b.c).max() for a, b in zip(fits, one))
print(f"max rel diff lambda {err:.3e}  c {errc:.3e}")
for n in (64, 128, 192, 256):
    xy = rng.uniform(0, 1, (n, 2)); y = np.sin(6 * xy[:, 0]) * np.cos(5 * xy[:, 1]) + 0.1 * rng.standard_normal(n)
    m.tps.fit_many([xy], [y])
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); m.tps.fit_many([xy], [y]); best = min(best, time.perf_counter() - t0)
    print(f"single batched fit n={n}: {best * 1e3:.3f} ms")
