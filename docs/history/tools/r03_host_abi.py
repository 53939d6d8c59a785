"""Round 3: the host-pointer ensemble entry point (what the R shim calls) against the resident one, float64 planes.
    python tools/r03_host_abi.py [side=10000] [bands ...]
Prints resident ms, then host-ABI ms for each MHS_HOST_BANDS value (0 = the library's own choice)."""
import ctypes as C
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import machisplin_amd as m
from machisplin_amd import synth, _lib

m.init()
side = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
bands = [int(a) for a in sys.argv[2:]] or [0, 1, 4, 8, 16]
g = synth.grid(side, side)
seed = synth.BASE_SEED + 3
planes, nodata = synth.covariates(g, 3, seed, dtype="f32")
xy, rows, cols, uv = synth.stations(g, 5000, seed)
cov_at = planes[:, torch.from_numpy(rows).cuda(), torch.from_numpy(cols).cuda()].cpu().numpy().astype(np.float64).T
X = np.column_stack([cov_at, xy])
y = synth.response(X, uv, seed)
params = synth.ensemble_params(X, y, seed)
models = [m.models.from_param_dict(p) for p in params]
wts = [0.2, 0.15, 0.1, 0.15, 0.2, 0.2]
stack64 = m.RasterStack(g, planes.to(torch.float64), nodata)
del planes
res = None
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = m.ensemble_predict(stack64, models, wts, 1.0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
print(f"resident f64: {dt:.1f} ms", flush=True)
# does a pageable host -> device copy running beside the kernels slow them down?  (the copy goes to a scratch buffer)
import threading
scratch = torch.empty_like(stack64.planes)
src_cpu = stack64.planes.cpu()
def _bg():
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        scratch.copy_(src_cpu, non_blocking=True)
    st.synchronize()
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = threading.Thread(target=_bg); th.start()
    res2 = m.ensemble_predict(stack64, models, wts, 1.0)
    torch.cuda.synchronize(); dt2 = (time.perf_counter() - t0) * 1e3
    th.join()
print(f"resident f64 with a 2.4 GB pageable upload running beside it: {dt2:.1f} ms (+{dt2 - dt:.1f})", flush=True)
del scratch, src_cpu, res2
host = np.ascontiguousarray(stack64.planes.cpu().numpy())
ref = res.cpu().numpy()
del stack64, res
torch.cuda.empty_cache()
out = np.empty((side, side))
hs = (C.c_void_p * len(models))(*[mm._h for mm in models])
ws = (C.c_double * len(models))(*wts)
st = _lib.Stack(host.ctypes.data, host.shape[0], _lib.F64, side * side, side, float("nan"))
gs = g.c_struct()
# resident, band by band on one stream: what cutting the grid into bands costs by itself
for nb in bands:
    if nb < 1: continue
    rp = (side + nb - 1) // nb
    full = torch.empty((side, side), dtype=torch.float64, device="cuda")
    stack_again = m.RasterStack(g, torch.from_numpy(host).cuda(), float("nan"))
    best = 1e9
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for b0 in range(0, side, rp):
            m.ensemble_predict(stack_again, models, wts, 1.0, window=(b0, min(side, b0 + rp), 0, side), out=full[b0:min(side, b0 + rp)])
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) * 1e3)
    print(f"resident f64 in {nb} bands, one stream: {best:.1f} ms (+{best - dt:.1f})", flush=True)
    del stack_again, full
    torch.cuda.empty_cache()
runs = [(nb, False, False) for nb in bands] + [(1, False, False)]
for nb, pageable, one in runs:
    if nb: os.environ["MHS_HOST_BANDS"] = str(nb)
    else: os.environ.pop("MHS_HOST_BANDS", None)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        _lib.check(_lib.lib().mhs_ensemble_predict(hs, ws, len(models), 1.0, C.byref(gs), C.byref(st), 0, side, 0, side, out.ctypes.data))
        best = min(best, (time.perf_counter() - t0) * 1e3)
    same = np.array_equal(np.nan_to_num(out), np.nan_to_num(ref))
    print(f"host ABI, bands {nb or 'auto (8 % | <= 50 M cells ... | 8 %)':>4}: {best:.1f} ms  (+{best - dt:.1f} over resident)  bitwise equal: {same}", flush=True)
os.environ.pop("MHS_HOST_BANDS", None)
os.environ["MHS_TIMING"] = "1"
_lib.check(_lib.lib().mhs_ensemble_predict(hs, ws, len(models), 1.0, C.byref(gs), C.byref(st), 0, side, 0, side, out.ctypes.data))
