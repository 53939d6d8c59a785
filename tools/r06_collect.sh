#!/bin/bash
# round 6: the evidence set for profiles/ (run on the GPU box through gpurun; results land in gpurun_out/r06_final/, copied to profiles/r06_*)
#   bash tools/r06_collect.sh [all|gputest|bench|benchmore|inlib|fit|pmc|fitpmc|forest|forestpmc|memberspmc|cfg5]
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_final
mkdir -p $O
export TMPDIR=/tmp
PART=${1:-all}
summ() { python - "$@" <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        print(f.split('/')[-1], {k:d.get(k) for k in ('value','ms_per_step','tps_fit_ms','reference_tiled_tps_ms','rsq_model','rsq_final')})
    except Exception as e:
        print(f, 'ERR', e)
PY
}
if [ "$PART" = all ] || [ "$PART" = gputest ]; then
  timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^/opt" | tail -25 > $O/gputest_full.log; tail -3 $O/gputest_full.log
fi
if [ "$PART" = all ] || [ "$PART" = bench ]; then
  timeout 900 python bench.py > $O/bench_cfg3_n1.json 2> $O/bench_cfg3_n1.err; tail -c 300 $O/bench_cfg3_n1.json; echo
  # the timed steps alone under rocprofv3 (the line's extra records -- float64 boundary, raster sensitivity -- launch the same
  # kernels on other windows and would blur the per-kernel statistics): the line's HIP-event launch times and the profiler's
  # durations of the full-grid dispatches from ONE run on ONE box
  # (MHS_RESERVE_KEEP=1: under the profiler, creating and destroying the CU-masked streams in every step -- what lifting the
  # reservation does since this round -- made every kernel of the run 3 x slower; the streams are kept for this run only)
  ( cd /tmp && rm -rf /tmp/kst && MHS_RESERVE_KEEP=1 MHS_BENCH_SKIP_F64=1 MHS_BENCH_SKIP_SENSITIVITY=1 MHS_BENCH_SKIP_FITTED=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o cfg3 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2>/tmp/kst.log )
  find /tmp/kst -name "*kernel_stats.csv" -exec cp {} $O/cfg3_steps_only_rocprofv3_kernel_stats.csv \;
  find /tmp/kst -name "*kernel_trace.csv" -exec cp {} /tmp/cfg3_kernel_trace.csv \;
  python tools/r06_launch_vs_rocprof.py $O/bench_under_rocprof.json $O/cfg3_steps_only_rocprofv3_kernel_stats.csv /tmp/cfg3_kernel_trace.csv > $O/cfg3_launch_vs_rocprof.json; cat $O/cfg3_launch_vs_rocprof.json
  summ $O/bench_cfg3_n1.json $O/bench_under_rocprof.json
fi
if [ "$PART" = all ] || [ "$PART" = benchmore ]; then
  timeout 900 python bench.py --tps-mode tiled > $O/bench_cfg3_n1_tiled_tps.json 2>/dev/null
  timeout 600 python bench.py --workload cfg2 > $O/bench_cfg2_n1.json 2>/dev/null
  timeout 600 python bench.py --workload cfg2 --tps-mode tiled --no-cpu-baseline > $O/bench_cfg2_n1_tiled_tps.json 2>/dev/null
  timeout 1200 python bench.py --workload cfg4 --steps 3 --warmup 1 > $O/bench_cfg4_n1.json 2>/dev/null
  MHS_BENCH_SKIP_F64=1 timeout 1800 python bench.py --workload cfg5 --steps 1 --warmup 1 > $O/bench_cfg5_n1.json 2>/dev/null
  summ $O/bench_cfg3_n1_tiled_tps.json $O/bench_cfg2_n1.json $O/bench_cfg2_n1_tiled_tps.json $O/bench_cfg4_n1.json $O/bench_cfg5_n1.json
fi
if [ "$PART" = all ] || [ "$PART" = inlib ]; then      # the library's own multi-device drivers: 1 slot, and 2 / 4 / 8 slots sharing this GPU (plumbing + overhead)
  for n in 1 2 4 8; do
    timeout 900 python bench.py --gpus $n --in-library --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cfg3_inlibrary_n$n.json 2> $O/bench_cfg3_inlibrary_n$n.err
  done
  for n in 1 4; do
    timeout 900 python bench.py --gpus $n --in-library --tps-mode tiled --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cfg3_inlibrary_tiled_n$n.json 2>/dev/null
  done
  timeout 1200 python bench.py --workload cfg4 --gpus 1 --in-library --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_cfg4_inlibrary_n1.json 2>/dev/null
  timeout 1200 python bench.py --workload cfg4 --gpus 4 --in-library --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_cfg4_inlibrary_n4.json 2>/dev/null
  python bench.py --gpus 2 --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_gpus2_no_launcher.out 2> $O/bench_gpus2_no_launcher.err; echo "exit code $?" >> $O/bench_gpus2_no_launcher.err
  summ $O/bench_cfg3_inlibrary_n*.json $O/bench_cfg3_inlibrary_tiled_n*.json $O/bench_cfg4_inlibrary_n*.json
fi
if [ "$PART" = all ] || [ "$PART" = fit ]; then
  timeout 600 python tools/fit_speed.py 500 2000 5000 10000 20000 2>&1 | grep -v "^/opt" > $O/fit_speed.txt; cat $O/fit_speed.txt
  MHS_TIMING=1 timeout 300 python tools/r06_batch_probe.py 2>&1 | grep -v "^/opt\|mhs_tps_fit n" > $O/batch_fit_probe.txt; tail -8 $O/batch_fit_probe.txt
  MHS_TIMING=1 timeout 300 python tools/tiled_speed.py 2>&1 | grep -v "^/opt\|mhs_tps_fit n" > $O/tiled_step3_4.txt; tail -8 $O/tiled_step3_4.txt
  bash tools/r06_fit_rocprof.sh 20000 5000 2000 > /dev/null; cp gpurun_out/r06_fit/fit_gcv_n*_kernel_stats.csv $O/
  ( cd /tmp && rm -rf /tmp/kstb && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstb -o tiled -- python $GRAFT_REPO_ROOT/tools/tiled_speed.py > /tmp/kstb.log 2>&1 )
  find /tmp/kstb -name "*kernel_stats.csv" -exec cp {} $O/tiled_step3_4_kernel_stats.csv \;
  head -8 $O/tiled_step3_4_kernel_stats.csv | cut -c1-160
fi
if [ "$PART" = all ] || [ "$PART" = pmc ]; then      # the batched fit kernel: instruction mix and LDS behaviour (separate passes, --pmc only)
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
    tag=$(echo $set | tr ' ' '_')
    ( cd /tmp && rm -rf /tmp/pmcb && timeout 600 rocprofv3 --pmc $set --output-format csv -d /tmp/pmcb -o b -- python $GRAFT_REPO_ROOT/tools/r06_batch_probe.py > /tmp/pmcb.log 2>&1 )
    find /tmp/pmcb -name "*counter_collection.csv" -exec cp {} $O/batch_fit_pmc_$tag.csv \;
  done
  python tools/r06_batch_pmc_summary.py $O > $O/batch_fit_pmc_summary.json; cat $O/batch_fit_pmc_summary.json
fi
if [ "$PART" = all ] || [ "$PART" = fitpmc ]; then      # MFMA counters of the large GCV fits (separate passes, --pmc only + --kernel-trace)
  for n in 20000 5000; do
    for set in "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES"; do
      tag=$(echo $set | tr ' ' '_')
      ( cd /tmp && rm -rf /tmp/pmcf && timeout 900 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmcf -o f -- python $GRAFT_REPO_ROOT/tools/fit_prof.py $n gcv 1 > /tmp/pmcf.log 2>&1 )
      find /tmp/pmcf -name "*counter_collection.csv" -exec cp {} /tmp/fit_pmc_${n}_$tag.csv \;
    done
    python tools/r06_fit_pmc_summary.py $n /tmp/fit_pmc_${n}_*.csv > $O/fit_gcv_n${n}_mfma_pmc.json; cat $O/fit_gcv_n${n}_mfma_pmc.json
  done
fi
if [ "$PART" = all ] || [ "$PART" = forest ]; then      # cfg5's forest: the block-subtree kernel against the whole-tree compact kernel, and what it does per block
  timeout 1100 python tools/r06_forest_big.py 60 20000 2>&1 | grep -v "^/opt" > $O/forest_big.txt; cat $O/forest_big.txt
  if [ -f machisplin_amd/libmhs_cbs_stats.so ]; then      # built by hand: forest.hip with -DRF_CBS_STATS (tools/README.md)
    MHS_LIB=machisplin_amd/libmhs_cbs_stats.so timeout 1100 python tools/r06_forest_stats.py 60 20000 2>&1 | grep -v "^/opt" > $O/forest_cbs_stats.txt; tail -5 $O/forest_cbs_stats.txt
  fi
  ( cd /tmp && rm -rf /tmp/kfo && MHS_BENCH_SKIP_F64=1 MHS_BENCH_SKIP_TILED=1 timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kfo -o cfg5 -- python $GRAFT_REPO_ROOT/bench.py --workload cfg5 --steps 1 --warmup 1 --no-cpu-baseline > /tmp/kfo.json 2>/tmp/kfo.log )
  find /tmp/kfo -name "*kernel_stats.csv" -exec cp {} $O/cfg5_rocprofv3_kernel_stats.csv \;
  head -6 $O/cfg5_rocprofv3_kernel_stats.csv | cut -c1-160
fi
if [ "$PART" = cfg5 ]; then
  MHS_BENCH_SKIP_F64=1 timeout 1800 python bench.py --workload cfg5 --steps 1 --warmup 1 > $O/bench_cfg5_n1.json 2>/dev/null
  summ $O/bench_cfg5_n1.json
fi
if [ "$PART" = forestpmc ]; then      # memory-side bytes of the two big-tree forest kernels (a --pmc pass of its own)
  ( cd /tmp && rm -rf /tmp/kfp && timeout 2000 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/kfp -o fb -- python $GRAFT_REPO_ROOT/tools/r06_forest_big.py 60 20000 > /tmp/kfp.log 2>&1 )
  tail -3 /tmp/kfp.log
  F=$(find /tmp/kfp -name "*counter_collection.csv" | head -1)
  python tools/r06_forest_pmc_summary.py $F 60 400000000 > $O/forest_fetch_size.json; head -40 $O/forest_fetch_size.json
fi
if [ "$PART" = memberspmc ]; then      # PMC passes on the member kernels of THIS build (bench.py quotes counters only for the build it runs): 8d planes, 1e8 cells
  bash tools/r04_members_pmc.sh 10000 brvs 8d > $O/members_pmc_8d.log 2>&1; tail -5 $O/members_pmc_8d.log
  cp gpurun_out/r4/pmc_8d/summary.json profiles/r06_8d_members_pmc_summary.json; cp gpurun_out/r4/pmc_8d/units.json profiles/r06_8d_members_pmc_units.json
  python tools/r04_pmc_derive.py r06_8d > /dev/null; cp profiles/r06_8d_members_pmc_*.json $O/; ls -la $O/r06_8d_members_pmc_*.json
fi
