set -x
mkdir -p gpurun_out/r06a
MHS_TIMING=1 python tools/small_fit_speed.py > gpurun_out/r06a/small_fit.txt 2>&1
MHS_TIMING=1 python tools/tiled_speed.py > gpurun_out/r06a/tiled.txt 2>&1
python tools/fit_speed.py > gpurun_out/r06a/fit_speed.txt 2>&1
MHS_TIMING=1 python - > gpurun_out/r06a/fit_phases.txt 2>&1 <<'PY'
import numpy as np, machisplin_amd as m, time
m.init()
for n in (2000, 5000):
    rng = np.random.default_rng(n)
    xy = rng.uniform(0, 1, (n, 2)); y = np.sin(6 * xy[:, 0]) * np.cos(5 * xy[:, 1]) + 0.1 * rng.standard_normal(n)
    m.Tps(xy, y); print("----", flush=True)
    m.Tps(xy, y)
    print("---- fixed", flush=True)
    m.Tps(xy, y, lam=1e-3)
PY
