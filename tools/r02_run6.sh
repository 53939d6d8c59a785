#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -m gpu -q -x -k "tps_fit" > gpurun_out/r02_gputest9.log 2>&1
tail -25 gpurun_out/r02_gputest9.log | cut -c1-200
for T in 6000 4000 3000 2000; do echo "== MHS_DELAY_T=$T"; MHS_DELAY_T=$T timeout 600 python tools/fit_speed.py 5000 10000 20000 2>&1 | grep GCV; done
