"""Round 6: does holding many CU-masked streams slow kernels on OTHER streams?  ksvm on 4 000^2 cells (torch's stream) fresh,
after mhs_fit_reserve_cus(32) + a tiled Step 3 through the batch (one plain stream), and after the same through the lanes
(MHS_TILES_BATCH=0: lanes 1..8 with their masked streams get created)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
import machisplin_amd as mhs
from machisplin_amd import synth, _lib
mhs.init(0)
side = 4000
g = synth.grid(side, side)
planes, nodata = synth.covariates(g, 3, 20251020, dtype="f32")
stack = mhs.RasterStack(g, planes, nodata)
xy, rows, cols, uv = synth.stations(g, 5000, 20251020)
cov = planes[:, torch.from_numpy(rows).cuda(), torch.from_numpy(cols).cuda()].cpu().numpy().astype(np.float64).T
X = np.column_stack([cov, xy])
resp = synth.response(X, uv, 20251020)
params = synth.ensemble_params(X, resp, 20251020, n_gbm_trees=200, n_rf_trees=10)
models = [mhs.models.from_param_dict(p) for p in params]
svr = models[[p["kind"] for p in params].index("svr")]
out = torch.empty((side, side), dtype=torch.float64, device="cuda")
def t_svr(tag):
    mhs.predict(stack, svr, out=out); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); mhs.predict(stack, svr, out=out); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"{tag}: ksvm {best*1e3*6.25:.1f} ms per 1e8 cells", flush=True)
t_svr("fresh")
prev = C.c_int(0)
resid = synth.tps_residual(uv, 3)
which = sys.argv[1] if len(sys.argv) > 1 else "onecall"
if "reserve" in which:
    _lib.check(_lib.lib().mhs_fit_reserve_cus(32, C.byref(prev)))
    _lib.check(_lib.lib().mhs_fit_reserve_cus(0, C.byref(prev)))
    t_svr("after mhs_fit_reserve_cus(32) then (0)")
g2 = synth.grid(6000, 6000)
xy2, _, _, uv2 = synth.stations(g2, 2500, 3)
res2 = synth.tps_residual(uv2, 3)
o2 = torch.empty((6000, 6000), dtype=torch.float64, device="cuda")
from machisplin_amd import tiles
nRx, nCx, fit_win, keep_win = tiles.step3_tile_windows(g2, 1500)
if "mosaic" in which:
    bufs = [torch.ones((int(k[1] - k[0]), int(k[3] - k[2])), dtype=torch.float64, device="cuda") for k in keep_win]
    tiles.mosaic_feather(g2, nRx, nCx, keep_win, bufs, merge_mode=False, out=o2); torch.cuda.synchronize()
    t_svr("after mosaic_feather alone")
if "evalone" in which:
    f = mhs.tps.fit_many([xy2[:200]], [res2[:200]])[0]
    mhs.interpolate(g2, f, out=o2); torch.cuda.synchronize()
    t_svr("after one far-field interpolate of a 200-knot spline on 6000^2")
if "composed" in which:
    mhs.tps_residual_surface(g2, xy2, res2, tile_edge=1500, out=o2, info={}); torch.cuda.synchronize()
    t_svr("after the Python-composed tiled Step 3 (fit_many + interpolate per tile + mosaic)")
if "rawmalloc" in which:
    hip = C.CDLL("libamdhip64.so")
    ptr = C.c_void_p()
    rc = hip.hipMalloc(C.byref(ptr), C.c_size_t(1 << 30)); torch.cuda.synchronize()
    t_svr("after a raw hipMalloc of 1 GB (rc %d)" % rc)
    rc = hip.hipMemset(ptr, 0, C.c_size_t(1 << 30)); torch.cuda.synchronize()
    t_svr("after hipMemset of it (rc %d)" % rc)
    hip.hipFree(ptr); torch.cuda.synchronize()
    t_svr("after hipFree")
if "mosaicfinite" in which:
    # the library's own arena + kernels on torch's stream, no fits: mhs_tps_surface_dev with every tile below 10 stations (zero tiles)
    g3 = synth.grid(6000, 6000)
    xy3, _, _, uv3 = synth.stations(g3, 40, 5)
    o3 = torch.empty((6000, 6000), dtype=torch.float64, device="cuda")
    mhs.tps_residual_surface(g3, xy3, synth.tps_residual(uv3, 5), tile_edge=1500, out=o3); torch.cuda.synchronize()
    t_svr("after a one-call tiled Step 3 whose tiles are all zero tiles (surface arena + memsets + mosaic only)")
if "otherstream" in which:
    f = mhs.tps.fit_many([xy2[:200]], [res2[:200]])[0]
    st2 = torch.cuda.Stream()
    mhs.interpolate(g2, f, out=o2, stream=st2.cuda_stream); torch.cuda.synchronize()
    t_svr("after one far-field interpolate on a SECOND torch stream")
    with torch.cuda.stream(st2):
        z = torch.randn(1 << 26, device="cuda", dtype=torch.float64); z = z * z + 1.0
    torch.cuda.synchronize()
    t_svr("after a torch elementwise kernel on that stream")
if "tiny" in which:
    g3 = synth.grid(600, 800)
    xy3, _, _, uv3 = synth.stations(g3, 1500, 5)
    o3 = torch.empty((600, 800), dtype=torch.float64, device="cuda")
    mhs.tps_residual_surface(g3, xy3, synth.tps_residual(uv3, 5), tile_edge=300, out=o3); torch.cuda.synchronize()
    t_svr("after a one-call tiled Step 3 on a 600 x 800 grid")
    _lib.check(_lib.lib().mhs_fit_reserve_cus(32, C.byref(prev)))
    _lib.check(_lib.lib().mhs_fit_reserve_cus(0, C.byref(prev)))
    t_svr("after another mhs_fit_reserve_cus(32) then (0)")
    mhs.predict(stack, models[0], out=out); torch.cuda.synchronize()
    t_svr("after another member's kernel")
if "lanes" in which:
    os.environ["MHS_TILES_BATCH"] = "0"
    mhs.tps_residual_surface(g2, xy2, res2, tile_edge=1500, out=o2); torch.cuda.synchronize()
    del os.environ["MHS_TILES_BATCH"]
    t_svr("after the one-call tiled Step 3 through the lanes (MHS_TILES_BATCH=0)")
if "onecall" in which:
    mhs.tps_residual_surface(g2, xy2, res2, tile_edge=1500, out=o2); torch.cuda.synchronize()
    t_svr("after the one-call tiled Step 3")
    for _ in range(3):
        mhs.tps_residual_surface(g2, xy2, res2, tile_edge=1500, out=o2)
    torch.cuda.synchronize()
    t_svr("after three more")
    big = torch.empty(1 << 28, dtype=torch.float64, device="cuda"); del big; torch.cuda.empty_cache(); torch.cuda.synchronize()
    t_svr("after a 2 GB torch allocation + empty_cache")
