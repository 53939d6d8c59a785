#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o g -- python $R/tools/fit_pmc.py gcv 20000 2 > /tmp/kt.log 2>&1
find /tmp/kt -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r02_gcv_n20000_kernel_stats.csv \;
cut -d, -f1-4 $R/gpurun_out/r02_gcv_n20000_kernel_stats.csv | head -16 | cut -c1-160
cd $R; MHS_FIT_TIMING=1 python tools/fit_pmc.py gcv 20000 1 2>&1 | grep "mhs_tps_fit\|gcv m"
