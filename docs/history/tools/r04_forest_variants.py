"""Round 4: stand-alone times of the randomForest (or, with MHS_TOOL_MEMBER=gbm, the gbm) kernel variants on cfg3's forest (5 000 stations, 500 trees) over side x side
cells of (a) the SURVEY 8d planes, (b) the bundled TWI / slope overviews mirrored to the window (+ synthetic alt), (c) 8d +
10 % white noise.  Variants are selected with the library's environment switches; every plane is compared bit for bit with the
first variant's.
    python tools/r04_forest_variants.py [side=8000] [reps=3] [name=value ...extra env for every variant]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import machisplin_amd as m  # noqa: E402
from machisplin_amd import synth  # noqa: E402

m.init()
side = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g10 = synth.grid(10000, 10000)
seed = synth.BASE_SEED + 3
xy, rows, cols, uv = synth.stations(g10, 5000, seed)
X = np.column_stack([synth.covariates_at(g10, 3, seed, rows, cols), xy])
y = synth.response(X, uv, seed)
member = os.environ.get("MHS_TOOL_MEMBER", "rf")
prm = synth.rf_params(X, y, seed) if member == "rf" else synth.gbm_params(X, y, seed)
mod = m.models.from_param_dict(prm)
g = synth.grid(side, side)
base, nodata = synth.covariates(g10, 3, seed, dtype="f32", window=(0, side, 0, side))


def rasters():
    yield "8d planes", base
    z = np.load(os.path.join(ROOT, "tests", "golden", "cfg1_extdata.npz"))
    real = base.clone()
    for k, name in ((1, "slope"), (2, "TWI")):
        a = torch.from_numpy(z[name].astype(np.float32)).cuda()
        a[a == float(z["nodata"])] = float("nan")
        t = torch.cat([a, a.flip(0)], 0)
        t = torch.cat([t, t.flip(1)], 1)
        reps_r, reps_c = -(-side // t.shape[0]), -(-side // t.shape[1])
        real[k] = t.repeat(reps_r, reps_c)[:side, :side]
    yield "bundled TWI / slope overviews (mirrored) + synthetic alt", real
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    noisy = base.clone()
    for k in range(3):
        lo, hi = synth.COV_RANGES[k]
        noisy[k] += 0.1 * (hi - lo) * (torch.rand((side, side), device="cuda", generator=gen) - 0.5)
    yield "8d + 10 % white noise", noisy


variants = [("one loader wave (even trees through its registers, odd trees by LDS-DMA) + 15 walking waves of 16 x 16 cells (round 4 default)", {}),
            ("the same with the LONG / LAT ranks searched like the covariates' (no per-column / per-row tables)", {"MHS_NO_AXIS_RANKS": "1"}),
            ("16 waves stage and walk (round 3: rf_walk_tb_kernel)", {"MHS_RF_NO_LOADER": "1"}),
            ("loader wave, walks start at the root", {"MHS_RF_NO_PREFIX": "1"}),
            ("two buffers + a barrier per tree, 5 walks (round 3's first half)", {"MHS_RF_DOUBLE_BUFFER": "1"}),
            ("one loader wave, every tree through its registers (no LDS-DMA for the odd trees)", {"MHS_RF_LD_FLAGS": "64"}),
            ("one loader wave, waves of 64 x 4 cells", {"MHS_RF_STRIP_WAVES": "1"}),
            ("two loader waves + 14 walking waves, waves of 16 x 16 cells", {"MHS_RF_TWO_LOADERS": "1"}),
            ("TIMING ONLY: keys and prefix only (no tree loop)", {"MHS_RF_LD_FLAGS": "128"}),
            ("TIMING ONLY: loader wave, walks off", {"MHS_RF_LD_FLAGS": "2"}),
            ("TIMING ONLY: loader wave, staging off", {"MHS_RF_LD_FLAGS": "4"}),
            ("TIMING ONLY: loader wave, walks + staging off", {"MHS_RF_LD_FLAGS": "6"}),
            ("TIMING ONLY: loader wave, the first 8 trees staged over and over (always in L2)", {"MHS_RF_LD_FLAGS": "32"}),
            ("TIMING ONLY: loader wave, walks off, the first 8 trees staged over and over", {"MHS_RF_LD_FLAGS": "34"})
]
if member == "gbm":
    variants = [("coherent kernel, waves of 16 x 16 cells, device-side probe (round 4 default)", {}),
                ("the same with the LONG / LAT ranks searched like the covariates'", {"MHS_NO_AXIS_RANKS": "1"}),
                ("coherent kernel, waves of 64 x 4 cells, probe (round 3)", {"MHS_GBM_STRIP_WAVES": "1"}),
                ("coherent kernel forced, 16 x 16", {"MHS_GBM_FORCE_COHERENT": "1"}),
                ("coherent kernel forced, 64 x 4", {"MHS_GBM_FORCE_COHERENT": "1", "MHS_GBM_STRIP_WAVES": "1"}),
                ("tree-order row-tile kernel", {"MHS_GBM_NO_COHERENT": "1"})]
out = torch.empty((side, side), dtype=torch.float64, device="cuda")
for rname, planes in rasters():
    stack = m.RasterStack(g, planes, nodata)
    ref = None
    for name, env in variants:
        for k, v in env.items():
            os.environ[k] = v
        m.predict(stack, mod, out=out)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            t0 = time.time()
            m.predict(stack, mod, out=out)
            torch.cuda.synchronize()
            best = min(best, time.time() - t0)
        for k in env:
            del os.environ[k]
        plane = out.clone()
        same = "" if ref is None else ("  == first" if torch.equal(torch.nan_to_num(plane), torch.nan_to_num(ref)) and torch.equal(torch.isnan(plane), torch.isnan(ref))
                                       else "  DIFFERS: max |diff| = %.3e (max |first| = %.3e)" % (float((torch.nan_to_num(plane) - torch.nan_to_num(ref)).abs().max()), float(torch.nan_to_num(ref).abs().max())))
        if ref is None:
            ref = plane
        print(f"[{rname}] {name:70s} {best * 1e3:9.2f} ms  -> 1e8 cells: {best * 1e8 / (side * side) * 1e3:8.1f} ms{same}", flush=True)
