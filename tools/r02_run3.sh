#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -q -x -k "tps_fit or golden" > gpurun_out/r02_gputest6.log 2>&1
tail -5 gpurun_out/r02_gputest6.log
timeout 900 python tools/fit_speed.py 2000 5000 10000 20000 > gpurun_out/r02_fit_speed_b.txt 2>&1
grep -v GCV gpurun_out/r02_fit_speed_b.txt
export TMPDIR=/tmp
cat > /tmp/fixed_only.py <<'PY'
import os, sys, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import machisplin_amd as m
m.init()
for n in (5000, 20000):
    rng = np.random.default_rng(n)
    xy = rng.uniform(0, 1, (n, 2)); y = np.sin(6*xy[:,0])*np.cos(5*xy[:,1]) + 0.1*rng.standard_normal(n)
    for _ in range(3): m.Tps(xy, y, lambda_=1e-3)
PY
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kfit -o fit -- python /tmp/fixed_only.py > /tmp/kfit.log 2>&1 )
find /tmp/kfit -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02_chol_kernel_stats_b.csv \;
head -14 gpurun_out/r02_chol_kernel_stats_b.csv | cut -c1-150
