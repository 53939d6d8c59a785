#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 0 32; do echo "RESERVE $r"; MHS_FIT_RESERVE_CUS=$r timeout 900 python bench.py --tps-mode tiled --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-260; done
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-230
timeout 900 python -m pytest tests/test_sharded_gpu.py tests/test_tps_eval_gpu.py tests/test_cfg1_gpu.py -m gpu -q -x 2>&1 | tail -3
