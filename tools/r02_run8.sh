#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_ensemble_gpu.py tests/test_full_size_gpu.py tests/test_sharded_gpu.py -m gpu -q -x 2>&1 | tail -5
timeout 900 python bench.py --workload cfg5 --steps 2 --warmup 1 2>&1 | tail -30 | cut -c1-2500
