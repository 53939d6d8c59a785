#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_tps_fit_gpu.py tests/test_tiles_gpu.py tests/test_sharded_gpu.py tests/test_cfg1_gpu.py -m gpu -q -x 2>&1 | tail -4
for i in 1 2; do timeout 1200 python bench.py --workload cfg4 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['unit_ms_mean_rank0'], d['unit_ms_max_rank0'], d['rsq_final_mean'])"; done
