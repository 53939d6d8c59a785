#!/bin/bash
cd $GRAFT_REPO_ROOT
for q in 4 16; do for r in 0 16; do
echo "HWQ $q RESERVE $r"
GPU_MAX_HW_QUEUES=$q MHS_FIT_TIMING=1 MHS_FIT_RESERVE_CUS=$r timeout 900 python bench.py --workload cfg3 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep "n=5000\] band\|^{" | cut -c1-220 | tail -6
done; done
