#!/bin/bash
# round 4: the evidence set for profiles/ (run on the GPU box through gpurun; results land in gpurun_out/r04_final/)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_final
mkdir -p $O
export TMPDIR=/tmp
PART=${1:-all}
if [ "$PART" = all ] || [ "$PART" = pmc ]; then
for RAST in 8d bundled; do
  bash tools/r04_members_pmc.sh 10000 brvs $RAST > $O/members_pmc_$RAST.log 2>&1
  cp gpurun_out/r4/pmc_$RAST/summary.json $O/${RAST}_members_pmc_summary.json; cp gpurun_out/r4/pmc_$RAST/units.json $O/${RAST}_members_pmc_units.json
  cp gpurun_out/r4/pmc_$RAST/summary.json profiles/r04_${RAST}_members_pmc_summary.json; cp gpurun_out/r4/pmc_$RAST/units.json profiles/r04_${RAST}_members_pmc_units.json
  python tools/r04_pmc_derive.py r04_$RAST > /dev/null; cp profiles/r04_${RAST}_members_pmc_derived.json $O/${RAST}_members_pmc_derived.json
done
fi
if [ "$PART" = all ] || [ "$PART" = bench ]; then
timeout 900 python bench.py > $O/bench_cfg3_n1.json 2> $O/bench_cfg3_n1.err; tail -c 300 $O/bench_cfg3_n1.json; echo
timeout 900 python bench.py --tps-mode tiled --no-cpu-baseline > $O/bench_cfg3_n1_tiled_tps.json 2>/dev/null
timeout 600 python bench.py --workload cfg2 > $O/bench_cfg2_n1.json 2>/dev/null
timeout 1200 python bench.py --workload cfg4 --steps 2 --warmup 1 > $O/bench_cfg4_n1.json 2>/dev/null
MHS_BENCH_SKIP_F64=1 timeout 1800 python bench.py --workload cfg5 --steps 1 --warmup 1 > $O/bench_cfg5_n1.json 2>/dev/null
( cd /tmp && rm -rf /tmp/kst && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o cfg3 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2>/tmp/kst.log )
find /tmp/kst -name "*kernel_stats.csv" -exec cp {} $O/cfg3_rocprofv3_kernel_stats.csv \;
find /tmp/kst -name "*domain_stats.csv" -exec cp {} $O/cfg3_rocprofv3_domain_stats.csv \;
head -8 $O/cfg3_rocprofv3_kernel_stats.csv | cut -c1-150
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print(sys.argv[1].split('/')[-1], {k:d.get(k) for k in ('value','ms_per_step','tps_fit_ms','rsq_model','rsq_final')})
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done
fi
if [ "$PART" = all ] || [ "$PART" = fit ]; then
timeout 600 python tools/fit_speed.py 500 2000 5000 10000 20000 2>&1 | grep -v "^/opt" > $O/fit_speed.txt; cat $O/fit_speed.txt
timeout 300 python tools/r04_fit_routes.py 2000 5000 2>&1 | grep -v "^/opt" > $O/fit_routes.txt; grep "GCV fit\|rel diff" $O/fit_routes.txt
cd /tmp
for cfg in "gcv 2000 3" "gcv 5000 3" "gcvcache 5000 3" "gcv 20000 1"; do
  set -- $cfg
  tag=${1}_n${2}
  rm -rf /tmp/kt_$tag
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$tag -o $tag -- python $GRAFT_REPO_ROOT/tools/fit_pmc.py $1 $2 $3 > /tmp/kt_$tag.log 2>&1
  echo "== kernel stats $tag rc=$?"
  find /tmp/kt_$tag -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/$O/fit_${tag}_kernel_stats.csv \;
  head -6 $GRAFT_REPO_ROOT/$O/fit_${tag}_kernel_stats.csv | cut -c1-150
done
for set in "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '_')
  for cfg in "gcv 5000" "gcv 20000"; do
    set2=($cfg)
    tag=${set2[0]}_n${set2[1]}
    rm -rf /tmp/pmc_$tag
    timeout 900 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- python $GRAFT_REPO_ROOT/tools/fit_pmc.py ${set2[0]} ${set2[1]} 1 > /tmp/pmc_$tag.log 2>&1
    echo "== pmc [$set] $tag rc=$?"; tail -1 /tmp/pmc_$tag.log
    find /tmp/pmc_$tag -name "*counter_collection.csv" -exec cp {} $GRAFT_REPO_ROOT/$O/fit_${tag}_pmc_${name}.csv \;
  done
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, collections, glob, json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r04_final"
out = {}
for f in sorted(glob.glob(O + "/fit_*_pmc_*.csv")):
    tag = os.path.basename(f).split("_pmc_")[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"].split("(")[0].split("::")[-1][:48]][row["Counter_Name"]] += float(row["Counter_Value"])
    for k, d in agg.items():
        out.setdefault(tag, {}).setdefault(k, {}).update(d)
json.dump(out, open(O + "/fit_pmc_summary.json", "w"), indent=1)
for tag, ks in out.items():
    print(tag)
    for k, d in sorted(ks.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU_MFMA_F64", 0))[:12]:
        print("   %-48s %s" % (k, {c: "%.3g" % v for c, v in d.items()}))
PY
rm -f $O/fit_*_pmc_*.csv
fi
if [ "$PART" = all ] || [ "$PART" = misc ]; then
timeout 900 python tools/r04_forest_variants.py 8000 3 2>&1 | grep -v "^/opt" > $O/forest_variants.txt; grep -v TIMING $O/forest_variants.txt | cut -c1-200
timeout 900 python tools/r03_tree_variants.py 6000 2 cfg5 2>&1 | grep -v "^/opt" > $O/tree_variants_cfg5.txt; cat $O/tree_variants_cfg5.txt
timeout 900 python tools/r03_host_abi.py 10000 0 1 2>&1 | grep -v "^/opt" > $O/host_abi.txt; cat $O/host_abi.txt
timeout 900 python tools/r03_host_abi.py 20000 0 1 2>&1 | grep -v "^/opt" > $O/host_abi_20000.txt; cat $O/host_abi_20000.txt
fi
