#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_ensemble_gpu.py tests/test_full_size_gpu.py tests/test_sharded_gpu.py -m gpu -q -x 2>&1 | tail -3
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('tps_fit_ms_overlapped_with_ensemble'))
for k in d.get('kernels', []): print(' ', k['kernel'], round(k['launch_ms'], 1))"
