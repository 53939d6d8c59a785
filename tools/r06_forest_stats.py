"""What rf_walk_cbs_kernel does per block on a cfg5-shaped forest (library built with -DRF_CBS_STATS; MHS_LIB=... selects it):
LDS slots staged, batches, trees entered at a terminal, levels walked, and the block's time by phase.
   MHS_LIB=machisplin_amd/libmhs_cbs_stats.so python tools/r06_forest_stats.py [trees] [rows] [stations] [columns] [covariates]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from machisplin_amd import _lib as L  # noqa: E402
if os.environ.get("MHS_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["MHS_LIB"])
import machisplin_amd as mhs  # noqa: E402
from machisplin_amd import synth  # noqa: E402

trees = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
stations = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
side = int(sys.argv[4]) if len(sys.argv) > 4 else 20000
ncov = int(sys.argv[5]) if len(sys.argv) > 5 else 5
os.environ["MHS_RF_KERNEL"] = "cbs"      # also for forests the loader-wave kernel would take
mhs.init()
geom = synth.grid(rows, side)
seed = synth.BASE_SEED + 3
planes, nodata = synth.covariates(geom, ncov, seed, dtype="f32")
xy, r, c, uv = synth.stations(geom, stations, seed)
cov = planes[:, torch.from_numpy(r).cuda(), torch.from_numpy(c).cuda()].cpu().numpy().astype(np.float64).T
X = np.column_stack([cov, xy])
y = synth.response(X, uv, seed)
prm = synth.rf_params(X, y, seed, n_trees=trees)
sizes = np.diff(prm["tree_offsets"])
print(f"forest: {trees} trees, nodes per tree {sizes.min()}..{sizes.max()}", flush=True)
model = mhs.models.from_param_dict(prm)
lib = L.lib()
fn = lib.mhs_debug_cbs_stats
fn.argtypes = [C.POINTER(C.c_ulonglong)]
buf = (C.c_ulonglong * 16)()
gen = torch.Generator(device="cuda")
gen.manual_seed(7)
variants = [("8d planes", planes)]
for frac in (0.01, 0.1):
    noisy = planes.clone()
    for k in range(ncov):
        lo, hi = synth.COV_RANGES[k % len(synth.COV_RANGES)]
        noisy[k] += (torch.rand((rows, side), device="cuda", generator=gen) - 0.5) * (frac * (hi - lo))
    variants.append(("8d + %g %% noise" % (100 * frac), noisy))
for vname, pl in variants:
    stack = mhs.RasterStack(geom, pl, nodata)
    out = mhs.predict(stack, model)
    torch.cuda.synchronize()
    fn(buf)
    out = mhs.predict(stack, model, out=out)
    torch.cuda.synchronize()
    assert fn(buf) == 0
    s = [int(v) for v in buf]
    blocks, waves = max(s[0], 1), max(s[6], 1)
    print(f"{vname}: blocks {blocks}, waves {waves}")
    print(f"   per block: LDS slots {s[1] / blocks:.0f} ({s[1] / blocks / trees:.1f} per tree), batches {s[2] / blocks:.2f}, trees entered at a terminal {s[3] / blocks:.1f} of {trees}")
    print(f"   per wave and tree: levels walked {s[4] / waves / trees:.2f}, wave entries at a terminal {s[5] / waves / trees:.3f}, levels above the wave's entry {s[7] / blocks / trees:.1f}")
    print(f"   the subtree below a WAVE's entry: {s[15] / waves / trees:.4f} of them above 400 LDS slots; the others {s[14] / max(waves * trees - s[15], 1):.1f} slots on average")
    cyc = [s[k] / waves for k in range(8, 14)]
    print("   cycles per wave: keys+ranges %.0f | block prefix %.0f | wave prefix %.0f | staging %.0f | walks %.0f | total %.0f" % tuple(cyc))
