#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ensemble_gpu.py -m gpu -q -x -k "ksvm or each_member" 2>&1 | tail -4
timeout 900 python bench.py --workload cfg3 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -3 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print(d['value'], d['ms_per_step']); print(d['roofline']['kernel'], d['roofline']['launch_ms'])
        for k in d.get('kernels', []): print(' ', k['kernel'], round(k['launch_ms'], 1), k.get('frac'))
"
