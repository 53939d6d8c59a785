#!/bin/bash
# round 2, GPU session 2: the MFMA Cholesky -- parity tests, speed, kernel trace
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x -k "tps_fit or golden or full_size and not cfg3" > gpurun_out/r02_gputest4.log 2>&1
tail -15 gpurun_out/r02_gputest4.log
timeout 900 python tools/fit_speed.py 500 2000 5000 10000 20000 > gpurun_out/r02_fit_speed_a.txt 2>&1
cat gpurun_out/r02_fit_speed_a.txt
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kfit -o fit -- python $GRAFT_REPO_ROOT/tools/fit_speed.py 5000 20000 > /tmp/kfit.log 2>&1 )
cp /tmp/kfit/fit_kernel_stats.csv gpurun_out/r02_fit_kernel_stats_a.csv 2>/dev/null || find /tmp/kfit -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02_fit_kernel_stats_a.csv \;
head -25 gpurun_out/r02_fit_kernel_stats_a.csv | cut -c1-170
python -m pytest tests/test_sharded_gpu.py -m gpu -q -x -k bench > gpurun_out/r02_gputest5.log 2>&1
tail -8 gpurun_out/r02_gputest5.log
