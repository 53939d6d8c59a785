"""Wall time of the device learner fits at the benchmark station counts, beside libsvm
(scikit-learn's SVR, one core) and the numpy vmmin of the oracle:   python tools/learn_fit_speed.py [n ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import machisplin_amd as mhs  # noqa: E402
from oracle import fit as of  # noqa: E402

mhs.init()
for n in [int(a) for a in sys.argv[1:]] or [5000, 20000]:
    p = 5 if n <= 5000 else 7
    rng = np.random.default_rng(n)
    X = rng.normal(size=(n, p))
    y = np.sin(2 * X[:, 0]) * np.cos(X[:, 1]) + 0.3 * X[:, 2] + 0.3 * rng.normal(size=n)
    sigma = 0.3
    t0 = time.perf_counter()
    m = mhs.models.Ksvm.fit(X, y, sigma)
    t1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    m = mhs.models.Ksvm.fit(X, y, sigma)
    t1 = min(t1, time.perf_counter() - t0)
    print(f"ksvm  n={n:6d} p={p}: {t1 * 1e3:9.1f} ms  {m.n_iter} SMO iterations ({t1 / m.n_iter * 1e6:.2f} us each), "
          f"{m.sv_index.size} support vectors", flush=True)
    try:
        from sklearn.svm import SVR
        Z = (X - X.mean(0)) / X.std(0, ddof=1)
        t = (y - y.mean()) / y.std(ddof=1)
        t0 = time.perf_counter()
        ref = SVR(kernel="rbf", gamma=sigma, C=1.0, epsilon=0.1, tol=1e-3, cache_size=4000).fit(Z, t)
        print(f"      libsvm (scikit-learn, 1 core): {(time.perf_counter() - t0) * 1e3:9.1f} ms, {ref.support_.size} support vectors", flush=True)
    except ImportError:
        pass
    w0 = rng.uniform(-0.7, 0.7, (p + 1) * 10 + 11)
    t0 = time.perf_counter()
    nn = mhs.models.Nnet.fit(X, y, w0)
    t1 = time.perf_counter() - t0
    print(f"nnet  n={n:6d} p={p}: {t1 * 1e3:9.1f} ms  {nn.counts[0]} function + {nn.counts[1]} gradient evaluations "
          f"({t1 / sum(nn.counts) * 1e6:.1f} us each), value {nn.value:.6g}, fail {nn.fail}", flush=True)
    if n <= 5000:
        ts = (y - y.min()) / (y - y.min()).max()
        t0 = time.perf_counter()
        w, val, nf, ng, fail = of.nnet_fit(X, ts, w0)
        print(f"      numpy vmmin (oracle): {(time.perf_counter() - t0) * 1e3:9.1f} ms  {nf} + {ng} evaluations, value {val:.6g}", flush=True)
