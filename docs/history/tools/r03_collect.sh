#!/bin/bash
# round 3: the evidence set for profiles/ (run on the GPU box through gpurun; results land in gpurun_out/r03_final/)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_final
mkdir -p $O
export TMPDIR=/tmp
bash tools/r03_members_pmc.sh 10000 brv > $O/members_pmc.log 2>&1; cp gpurun_out/r3/pmc/summary.json $O/members_pmc_summary.json; cp gpurun_out/r3/pmc/units.json $O/members_pmc_units.json; cp gpurun_out/r3/pmc/summary.json profiles/r03_members_pmc_summary.json; cp gpurun_out/r3/pmc/units.json profiles/r03_members_pmc_units.json; python tools/r03_pmc_derive.py > /dev/null; cp profiles/r03_members_pmc_derived.json $O/members_pmc_derived.json
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/gputest_full.log 2>&1; tail -4 $O/gputest_full.log | cut -c1-160
timeout 900 python bench.py > $O/bench_cfg3_n1.json 2> $O/bench_cfg3_n1.err; tail -c 300 $O/bench_cfg3_n1.json; echo
timeout 900 python bench.py --tps-mode tiled --no-cpu-baseline > $O/bench_cfg3_n1_tiled_tps.json 2>/dev/null
timeout 600 python bench.py --workload cfg2 > $O/bench_cfg2_n1.json 2>/dev/null
timeout 1200 python bench.py --workload cfg4 --steps 2 --warmup 1 > $O/bench_cfg4_n1.json 2>/dev/null
MHS_BENCH_SKIP_F64=1 timeout 1800 python bench.py --workload cfg5 --steps 1 --warmup 1 > $O/bench_cfg5_n1.json 2>/dev/null
( cd /tmp && rm -rf /tmp/kst && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o cfg3 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2>/tmp/kst.log )
find /tmp/kst -name "*kernel_stats.csv" -exec cp {} $O/cfg3_rocprofv3_kernel_stats.csv \;
find /tmp/kst -name "*domain_stats.csv" -exec cp {} $O/cfg3_rocprofv3_domain_stats.csv \;
head -8 $O/cfg3_rocprofv3_kernel_stats.csv | cut -c1-150
( cd /tmp && rm -rf /tmp/kst4 && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst4 -o cfg4 -- python $GRAFT_REPO_ROOT/bench.py --workload cfg4 --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_cfg4_under_rocprof.json 2>/tmp/kst4.log )
find /tmp/kst4 -name "*kernel_stats.csv" -exec cp {} $O/cfg4_rocprofv3_kernel_stats.csv \;
timeout 600 python tools/fit_speed.py 500 2000 5000 10000 20000 2>&1 | grep -v "^/opt" > $O/fit_speed.txt; cat $O/fit_speed.txt
timeout 900 python tools/r03_tree_variants.py 8000 3 2>&1 | grep -v "^/opt" > $O/tree_variants.txt; cat $O/tree_variants.txt
timeout 900 python tools/r03_tree_variants.py 6000 2 cfg5 2>&1 | grep -v "^/opt" > $O/tree_variants_cfg5.txt; cat $O/tree_variants_cfg5.txt
timeout 900 python tools/r03_gbm_coherent.py 8000 3 2>&1 | grep -v "^/opt" > $O/gbm_coherent.txt; cat $O/gbm_coherent.txt
timeout 900 python tools/r03_host_abi.py 10000 0 1 2>&1 | grep -v "^/opt" > $O/host_abi.txt; cat $O/host_abi.txt
timeout 900 python tools/r03_host_abi.py 20000 0 1 2>&1 | grep -v "^/opt" > $O/host_abi_20000.txt; cat $O/host_abi_20000.txt
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print(sys.argv[1].split('/')[-1], {k:d.get(k) for k in ('value','ms_per_step','tps_fit_ms','rsq_model','rsq_final')})
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done
