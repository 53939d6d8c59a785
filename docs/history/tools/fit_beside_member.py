"""Does mhs_fit_reserve_cus let a Tps fit run beside a grid-filling ensemble member?  One member on a big window on the
null stream, the n = 5 000 GCV fit started right after it, with 0 / 8 / 16 compute units reserved.
   python tools/fit_beside_member.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import machisplin_amd as mhs  # noqa: E402
from machisplin_amd import models, synth  # noqa: E402

mhs.init()
side = 8000
geom = synth.grid(side, side)
seed = synth.BASE_SEED + 3
planes, nodata = synth.covariates(geom, 3, seed, dtype="f32")
stack = mhs.RasterStack(geom, planes, nodata)
xy, r, c, uv = synth.stations(geom, 5000, seed)
cov = planes[:, torch.from_numpy(r).cuda(), torch.from_numpy(c).cuda()].cpu().numpy().astype(np.float64).T
X = np.column_stack([cov, xy])
y = synth.response(X, uv, seed)
prm = synth.ensemble_params(X, y, seed, n_gbm_trees=10000, n_rf_trees=500)
mods = {p["kind"]: models.from_param_dict(p) for p in prm}
res = synth.tps_residual(uv, seed)
out = torch.zeros((side, side), dtype=torch.float64, device="cuda")
mhs.Tps(xy, res)
t0 = time.perf_counter(); mhs.Tps(xy, res); print(f"fit alone {1e3 * (time.perf_counter() - t0):.1f} ms")
for reserve in (16, 32):
    models.fit_reserve_cus(reserve)
    mhs.Tps(xy, res)
    t0 = time.perf_counter(); mhs.Tps(xy, res); print(f"fit alone, confined to {reserve} CUs: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
side_stream = torch.cuda.Stream()          # non-blocking: CU-masked streams are blocking ones, i.e. they synchronise with the NULL stream
torch.cuda.set_stream(side_stream)
for kind in ("gbm", "rf", "svr"):
    for reserve in (0, 16, 32):
        models.fit_reserve_cus(reserve)
        models.members_predict(stack, [mods[kind]], [1.0], out=out)     # warm-up (tables)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        models.members_predict(stack, [mods[kind]], [1.0], out=out)
        e1.record()
        t1 = time.perf_counter()
        fit = mhs.Tps(xy, res)
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        print(f"{kind:4s} reserve {reserve:2d}: launch {1e3 * (t1 - t0):6.1f} ms, fit returned after {1e3 * (t2 - t0):7.1f} ms, "
              f"member done after {1e3 * (t3 - t0):7.1f} ms (member on its stream: {e0.elapsed_time(e1):.1f} ms)", flush=True)
models.fit_reserve_cus(0)
