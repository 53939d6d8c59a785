#!/bin/bash
# round 5: the evidence set for profiles/ (run on the GPU box through gpurun; results land in gpurun_out/r05_final/ and profiles/r05_*)
#   bash tools/r05_collect.sh [all|pmc|pmcsub|bench|benchmore|inlib|fit|misc]
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_final
mkdir -p $O
export TMPDIR=/tmp
PART=${1:-all}
pmc_set() {   # $1 = raster set, $2 = tag suffix
  bash tools/r04_members_pmc.sh 10000 brvs $1 > $O/members_pmc_$2.log 2>&1
  cp gpurun_out/r4/pmc_$1/summary.json profiles/r05_$2_members_pmc_summary.json; cp gpurun_out/r4/pmc_$1/units.json profiles/r05_$2_members_pmc_units.json
  python tools/r04_pmc_derive.py r05_$2 > /dev/null; cp profiles/r05_$2_members_pmc_*.json $O/
}
if [ "$PART" = all ] || [ "$PART" = pmc ]; then
  pmc_set 8d 8d
  pmc_set bundled bundled
fi
if [ "$PART" = all ] || [ "$PART" = pmcsub ]; then      # the forest's subtree-staging kernel: its traffic beside the default's
  MHS_RF_KERNEL=sub bash tools/r04_members_pmc.sh 10000 r 8d > $O/members_pmc_8d_rfsub.log 2>&1
  cp gpurun_out/r4/pmc_8d/summary.json profiles/r05_8d_rfsub_members_pmc_summary.json; cp gpurun_out/r4/pmc_8d/units.json profiles/r05_8d_rfsub_members_pmc_units.json
  python tools/r04_pmc_derive.py r05_8d_rfsub > /dev/null; cp profiles/r05_8d_rfsub_members_pmc_*.json $O/
fi
summ() { python - "$@" <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        print(f.split('/')[-1], {k:d.get(k) for k in ('value','ms_per_step','tps_fit_ms','rsq_model','rsq_final')})
    except Exception as e:
        print(f, 'ERR', e)
PY
}
if [ "$PART" = all ] || [ "$PART" = bench ]; then
  timeout 900 python bench.py > $O/bench_cfg3_n1.json 2> $O/bench_cfg3_n1.err; tail -c 300 $O/bench_cfg3_n1.json; echo
  ( cd /tmp && rm -rf /tmp/kst && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o cfg3 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2>/tmp/kst.log )
  find /tmp/kst -name "*kernel_stats.csv" -exec cp {} $O/cfg3_rocprofv3_kernel_stats.csv \;
  find /tmp/kst -name "*domain_stats.csv" -exec cp {} $O/cfg3_rocprofv3_domain_stats.csv \;
  head -8 $O/cfg3_rocprofv3_kernel_stats.csv | cut -c1-150
  summ $O/bench_cfg3_n1.json $O/bench_under_rocprof.json
fi
if [ "$PART" = all ] || [ "$PART" = benchmore ]; then
  timeout 900 python bench.py --tps-mode tiled --no-cpu-baseline > $O/bench_cfg3_n1_tiled_tps.json 2>/dev/null
  timeout 600 python bench.py --workload cfg2 > $O/bench_cfg2_n1.json 2>/dev/null
  timeout 1200 python bench.py --workload cfg4 --steps 2 --warmup 1 > $O/bench_cfg4_n1.json 2>/dev/null
  MHS_BENCH_SKIP_F64=1 timeout 1800 python bench.py --workload cfg5 --steps 1 --warmup 1 > $O/bench_cfg5_n1.json 2>/dev/null
  summ $O/bench_cfg3_n1_tiled_tps.json $O/bench_cfg2_n1.json $O/bench_cfg4_n1.json $O/bench_cfg5_n1.json
fi
if [ "$PART" = all ] || [ "$PART" = inlib ]; then      # the library's own multi-device drivers: 1 slot, and 2 / 4 / 8 slots sharing this GPU (plumbing + overhead)
  for n in 1 2 4 8; do
    timeout 900 python bench.py --gpus $n --in-library --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cfg3_inlibrary_n$n.json 2> $O/bench_cfg3_inlibrary_n$n.err
  done
  timeout 900 python bench.py --gpus 4 --in-library --tps-mode tiled --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cfg3_inlibrary_tiled_n4.json 2>/dev/null
  timeout 1200 python bench.py --workload cfg4 --gpus 1 --in-library --steps 2 --warmup 1 > $O/bench_cfg4_inlibrary_n1.json 2>/dev/null
  timeout 1200 python bench.py --workload cfg4 --gpus 4 --in-library --steps 2 --warmup 1 > $O/bench_cfg4_inlibrary_n4.json 2>/dev/null
  summ $O/bench_cfg3_inlibrary_n*.json $O/bench_cfg3_inlibrary_tiled_n4.json $O/bench_cfg4_inlibrary_n*.json
fi
if [ "$PART" = all ] || [ "$PART" = fit ]; then
  timeout 600 python tools/fit_speed.py 500 2000 5000 10000 20000 2>&1 | grep -v "^/opt" > $O/fit_speed.txt; cat $O/fit_speed.txt
fi
if [ "$PART" = all ] || [ "$PART" = misc ]; then
  timeout 900 python tools/r05_forest_ring.py 8000 2>&1 | grep -v "^/opt" > $O/forest_kernels.txt; cat $O/forest_kernels.txt
  timeout 900 python tools/r03_host_abi.py 10000 0 1 2>&1 | grep -v "^/opt" > $O/host_abi.txt; cat $O/host_abi.txt
fi
