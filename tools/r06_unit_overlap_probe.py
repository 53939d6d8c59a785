"""Does a batch of small spline fits (the reference-tiled Step 3 of a cfg4 unit: 16 tiles of ~80-250 stations, one workgroup
each, ~110 KB of LDS) run BESIDE the unit's ksvm grid kernel, or behind it?  The ksvm kernel (25 M cells x 757 support
vectors) is launched on torch's stream and mhs_tps_fit_many right behind it on the library's own stream; wall time of the fits
alone, of the grid kernel alone, and of both.
   python tools/r06_unit_overlap_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import machisplin_amd as mhs  # noqa: E402
from machisplin_amd import synth  # noqa: E402

mhs.init()
side, layers = 5025, 3
g = synth.grid(side, side)
seed = synth.BASE_SEED + 4
planes, nodata = synth.covariates(g, layers, seed, dtype="f32")
stack = mhs.RasterStack(g, planes, nodata)
xy, rows, cols, uv = synth.stations(g, 1262, seed)
cov = planes[:, torch.from_numpy(rows).cuda(), torch.from_numpy(cols).cuda()].cpu().numpy().astype(np.float64).T
X = np.column_stack([cov, xy])
y = synth.response(X, uv, seed)
params = synth.ensemble_params(X, y, seed, which="gnmv")
svr = mhs.models.from_param_dict(next(p for p in params if p["kind"] == "svr"))
rng = np.random.default_rng(3)
sets = []
for k in range(16):
    idx = rng.choice(1262, size=int(rng.integers(80, 250)), replace=False)
    sets.append((xy[idx], y[idx] - y[idx].mean()))
xs, ys = [a for a, _ in sets], [b for _, b in sets]
out = torch.empty((side, side), dtype=torch.float64, device="cuda")


def fits():
    t0 = time.perf_counter()
    mhs.tps.fit_many(xs, ys)
    return (time.perf_counter() - t0) * 1e3


def grid():
    t0 = time.perf_counter()
    mhs.predict(stack, svr, out=out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for _ in range(2):
    fits(); grid()
print("fits alone      : %.2f ms" % min(fits() for _ in range(5)))
print("grid alone      : %.2f ms" % min(grid() for _ in range(5)))
for label, first in (("grid launched first", True), ("fits launched first", False)):
    best = None
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if first:
            mhs.predict(stack, svr, out=out)          # asynchronous launch on torch's stream
            tf = fits()                                 # the library's own stream; returns when the fits are done
        else:
            import threading
            box = {}
            th = threading.Thread(target=lambda: box.setdefault("t", fits()))
            th.start()
            time.sleep(0.0006)                          # let the fit kernel reach the GPU first
            mhs.predict(stack, svr, out=out)
            th.join()
            tf = box["t"]
        torch.cuda.synchronize()
        tot = (time.perf_counter() - t0) * 1e3
        if best is None or tot < best[0]:
            best = (tot, tf)
    print("%s: both %.2f ms, the fits' call returned after %.2f ms" % (label, best[0], best[1]))
