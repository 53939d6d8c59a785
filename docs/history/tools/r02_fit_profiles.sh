#!/bin/bash
# round 2: rocprofv3 evidence for the spline fit (run on the GPU box through gpurun): kernel-trace statistics and PMC
# passes (MFMA instruction / busy counters, VALU, HBM traffic) for the fixed-lambda (MFMA Cholesky) and GCV routes.
# PMC passes are separate runs with --kernel-trace only, as the pool requires.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_prof
mkdir -p $O
rocprofv3 -L > $O/counters_available.txt 2>&1
grep -i -E "MFMA|FETCH_SIZE|WRITE_SIZE|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES|GRBM_GUI_ACTIVE" $O/counters_available.txt | head -40
for cfg in "fixed 20000" "fixed 5000" "gcv 5000"; do
  set -- $cfg
  tag=${1}_n${2}
  rm -rf /tmp/kt_$tag
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$tag -o $tag -- python $R/tools/fit_pmc.py $1 $2 3 > /tmp/kt_$tag.log 2>&1
  echo "== kernel stats $tag rc=$?"
  find /tmp/kt_$tag -name "*kernel_stats.csv" -exec cp {} $O/${tag}_kernel_stats.csv \;
  head -8 $O/${tag}_kernel_stats.csv | cut -c1-150
done
for set in "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '_')
  for cfg in "fixed 20000" "gcv 5000"; do
    set2=($cfg)
    tag=${set2[0]}_n${set2[1]}
    rm -rf /tmp/pmc_$tag
    timeout 900 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- python $R/tools/fit_pmc.py ${set2[0]} ${set2[1]} 1 > /tmp/pmc_$tag.log 2>&1
    echo "== pmc [$set] $tag rc=$?"; tail -2 /tmp/pmc_$tag.log
    find /tmp/pmc_$tag -name "*counter_collection.csv" -exec cp {} $O/${tag}_pmc_${name}.csv \;
  done
done
python3 - <<'PY'
import collections, csv, glob, json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r02_prof"
out = {}
for f in sorted(glob.glob(O + "/*_pmc_*.csv")):
    tag = os.path.basename(f).split("_pmc_")[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(int)
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(k, row["Counter_Name"])] += 1
    for k, d in agg.items():
        for c, v in d.items():
            out.setdefault(tag, {}).setdefault(k, {})[c] = {"sum": v, "dispatches": cnt[(k, c)]}
json.dump(out, open(O + "/pmc_summary.json", "w"), indent=1)
for tag, ks in out.items():
    for k, d in ks.items():
        if "syrk" in k or "trsm" in k or "band_symm" in k or "band_update" in k:
            print(tag, k[:40], {c: (round(v["sum"]), v["dispatches"]) for c, v in d.items()})
PY
