"""Do two grids share the chip?  Each big ensemble member on a normal-priority stream, a second kernel on another
stream (normal / high priority) 5 ms later: total times, and when the second kernel finishes.  Evidence for DESIGN.md
section 9 (members do not overlap; a prioritised small grid gets in after 1-3 ms; a prioritised big one takes over)."""
import time, numpy as np, torch, sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import machisplin_amd as m
from machisplin_amd import synth
m.init()
side = 6000
g = synth.grid(side, side)
seed = synth.BASE_SEED + 3
planes, nodata = synth.covariates(g, 3, seed, dtype="f32")
stack = m.RasterStack(g, planes, nodata)
xy, rows, cols, uv = synth.stations(g, 5000, seed)
cov_at = planes[:, torch.from_numpy(rows).cuda(), torch.from_numpy(cols).cuda()].cpu().numpy().astype(np.float64).T
X = np.column_stack([cov_at, xy])
y = synth.response(X, uv, seed)
params = synth.ensemble_params(X, y, seed)
mods = {p["kind"]: m.models.from_param_dict(p) for p in params}
o1 = torch.empty((side, side), dtype=torch.float64, device="cuda"); o2 = torch.empty_like(o1)
lo, hi = torch.cuda.Stream(priority=0), torch.cuda.Stream(priority=-1)
def run(a, b, sa, sb):
    torch.cuda.synchronize(); t0 = time.time()
    m.predict(stack, mods[a], out=o1, stream=sa.cuda_stream)
    time.sleep(0.005)      # let the first grid occupy the chip
    m.predict(stack, mods[b], out=o2, stream=sb.cuda_stream)
    torch.cuda.synchronize(); return (time.time() - t0) * 1e3
for k in ("rf", "svr", "gbm", "nnet"):
    m.predict(stack, mods[k], out=o1); torch.cuda.synchronize()
    t0 = time.time(); m.predict(stack, mods[k], out=o1); torch.cuda.synchronize(); print(k, "alone", round((time.time() - t0) * 1e3, 1))
for a, b in (("rf", "gbm"), ("rf", "svr"), ("gbm", "rf"), ("gbm", "nnet")):
    run(a, b, lo, hi)
    print(a, "(normal) then", b, "(high priority): same stream", round(run(a, b, lo, lo), 1), " two streams", round(run(a, b, lo, hi), 1))
def when(a, b, sa, sb):
    torch.cuda.synchronize(); t0 = time.time()
    m.predict(stack, mods[a], out=o1, stream=sa.cuda_stream)
    time.sleep(0.005)
    m.predict(stack, mods[b], out=o2, stream=sb.cuda_stream)
    ev = torch.cuda.Event(); ev.record(sb)
    ev.synchronize(); tb = (time.time() - t0) * 1e3
    torch.cuda.synchronize(); return tb, (time.time() - t0) * 1e3
for a, b in (("gbm", "nnet"), ("rf", "nnet"), ("svr", "nnet"), ("rf", "svr")):
    when(a, b, lo, hi)
    tb, tt = when(a, b, lo, hi)
    print(a, "then", b, "(high priority): second done at", round(tb, 1), "ms, all done at", round(tt, 1))
