#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/k5; MHS_BENCH_SKIP_F64=1 timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k5 -o cfg5 -- python $R/bench.py --workload cfg5 --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r02_bench_cfg5_under_rocprof.json 2>/tmp/k5.log
find /tmp/k5 -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r02_cfg5_rocprofv3_kernel_stats.csv \;
head -12 $R/gpurun_out/r02_cfg5_rocprofv3_kernel_stats.csv | cut -c1-170
cd $R; timeout 600 python tools/learn_fit_speed.py 5000 20000 2>&1 | grep -v "^/opt" > gpurun_out/r02_learn_fit_speed.txt; cat gpurun_out/r02_learn_fit_speed.txt
