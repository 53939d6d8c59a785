#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kst; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o cfg3 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2>/tmp/kst.log; echo "rocprofv3 exit $?"
find /tmp/kst -name "*kernel_stats.csv" -exec cp {} $O/cfg3_rocprofv3_kernel_stats.csv \;
find /tmp/kst -name "*domain_stats.csv" -exec cp {} $O/cfg3_rocprofv3_domain_stats.csv \;
head -7 $O/cfg3_rocprofv3_kernel_stats.csv | cut -c1-60,200-330
cd $R
timeout 1200 python bench.py --workload cfg4 --steps 3 --warmup 1 > $O/bench_cfg4_n1.json 2>/dev/null; tail -c 400 $O/bench_cfg4_n1.json; echo
timeout 900 python -m pytest tests -m gpu -q > $O/gputest_full.log 2>&1; tail -2 $O/gputest_full.log
