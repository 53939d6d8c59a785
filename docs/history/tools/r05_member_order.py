import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import machisplin_amd as m
from machisplin_amd import synth
m.init()
side = 10000
g = synth.grid(side, side)
seed = synth.BASE_SEED + 3
planes, nodata = synth.covariates(g, 3, seed, dtype="f32")
xy, rows, cols, uv = synth.stations(g, 5000, seed)
cov = planes[:, torch.from_numpy(rows).cuda(), torch.from_numpy(cols).cuda()].cpu().numpy().astype(np.float64).T
X = np.column_stack([cov, xy])
y = synth.response(X, uv, seed)
prm = synth.ensemble_params(X, y, seed)
kinds = [p["kind"] for p in prm]
models = [m.models.from_param_dict(p) for p in prm]
_, wts, tot = m.models.select_weights(synth.OPTX_WEIGHTS)
stack = m.RasterStack(g, planes, nodata)
torch.cuda.set_stream(torch.cuda.Stream())
def run(order, label):
    ms = [models[i] for i in order]; ws = [wts[i] for i in order]
    for _ in range(2): m.mltps_predict(stack, xy, y, ms, ws, tot, tile_edge=None); torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = m.mltps_predict(stack, xy, y, ms, ws, tot, tile_edge=None); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(label, [kinds[i] for i in order], "step ms:", np.round(ts, 1), flush=True)
run([0, 1, 2, 3, 4, 5], "reference order ")
run([0, 1, 2, 3, 5, 4], "svr before rf    ")
run([4, 0, 1, 2, 3, 5], "rf first         ")
run([5, 0, 1, 2, 3, 4], "svr first        ")
run([0, 1, 2, 3, 4, 5], "reference order ")
