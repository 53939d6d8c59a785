"""Per-phase times of the batched fit's step loop (library built with -DSB_PHASE_TRACE): MHS_LIB=exp/libtraceN.so python tools/r06_batch_trace.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import machisplin_amd._lib as L
if os.environ.get("MHS_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["MHS_LIB"])
import machisplin_amd as m
m.init()
rng = np.random.default_rng(1)
for n in (128, 192, 256):
    xy = rng.uniform(0, 1, (n, 2)); y = np.sin(6 * xy[:, 0]) * np.cos(5 * xy[:, 1]) + 0.1 * rng.standard_normal(n)
    m.tps.fit_many([xy], [y]); print(f"n={n}", file=sys.stderr, flush=True)
    m.tps.fit_many([xy], [y])
