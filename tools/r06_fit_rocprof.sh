#!/bin/bash
# rocprofv3 kernel statistics of GCV fits at the given sizes: bash tools/r06_fit_rocprof.sh 20000 5000 -> gpurun_out/r06_fit/
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06_fit; mkdir -p $O
for n in "$@"; do
  ( cd /tmp && rm -rf /tmp/kst_$n && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst_$n -o fit -- python $GRAFT_REPO_ROOT/tools/fit_prof.py $n gcv > /tmp/kst_$n.log 2>&1 )
  find /tmp/kst_$n -name "*kernel_stats.csv" -exec cp {} $O/fit_gcv_n${n}_kernel_stats.csv \;
  head -12 $O/fit_gcv_n${n}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-140
done
