"""Phase stamps of band_panel_reg_kernel.  Needs a library built with -DMHS_PANEL_TRACE (tps_fit.hip):
   make -C machisplin_amd/csrc CXXFLAGS+=-DMHS_PANEL_TRACE   (then rebuild without it)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import machisplin_amd as m
from machisplin_amd import _lib
m.init()
n = 5000
rng = np.random.default_rng(n)
xy = rng.uniform(0, 1, (n, 2)); y = np.sin(6 * xy[:, 0]) * np.cos(5 * xy[:, 1]) + 0.1 * rng.standard_normal(n)
for _ in range(3):
    t = m.Tps(xy, y)
if not hasattr(_lib.load(), "mhs_debug_panel_trace"):
    sys.exit("this libmachisplin_hip.so was built without -DMHS_PANEL_TRACE")
buf = (C.c_ulonglong * 64)()
fn = _lib.load().mhs_debug_panel_trace
fn.argtypes = [C.c_void_p]; fn.restype = C.c_int
assert fn(buf) == 0
names = ["start", "loaded", "norms"] + [f"step {j}" for j in range(8)] + ["steps done", "V'g reduced", "z formed", "stored + T"]
inner = ["entry", "local products", "lane-swap reduce + publish", "norm, sqrt, divisions", "barrier", "totals + readlane", "update"]
for slot, what in ((0, "first panel (t = 4989)"), (1, "panel at t ~ 2500")):
    v = list(buf[slot * 32:slot * 32 + 32])
    print(what, ": total cycles", v[14] - v[0])
    for k in range(1, 15):
        print(f"  {names[k]:14s} +{v[k] - v[k - 1]}")
    prev = v[6]
    print("  inside step 3:")
    for k, nm in zip(range(15, 22), inner):
        print(f"    {nm:28s} +{v[k] - prev}"); prev = v[k]
    print(f"    {'rotation':28s} +{v[7] - prev}")
