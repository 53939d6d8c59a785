#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -m gpu -q -x -k "tps_fit or cfg5_gcv or golden" > gpurun_out/r02_gputest10.log 2>&1
tail -8 gpurun_out/r02_gputest10.log | cut -c1-200
timeout 600 python tools/fit_speed.py 5000 5400 10000 20000 2>&1 | grep -v "^/opt" > gpurun_out/r02_fit_speed_e.txt
cat gpurun_out/r02_fit_speed_e.txt
MHS_FIT_TIMING=1 python tools/fit_pmc.py gcv 20000 1 2>&1 | grep "mhs_tps_fit\|gcv m"
