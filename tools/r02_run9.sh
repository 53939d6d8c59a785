#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 0 1; do
if [ $v = 1 ]; then export MHS_NULL_STREAM_COPIES=1; fi
echo "NULL_STREAM_COPIES=$v"
timeout 1200 python bench.py --workload cfg4 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print({k: d[k] for k in d if 'unit' in k or 'ms' in k})
"
done
