#!/bin/bash
# round 3: rocprofv3 evidence for the spline fit's GCV route (run on the GPU box through gpurun): kernel-trace statistics of the
# full fit at n = 5 000 / 20 000 and of refits through the band-route reduction cache, and the MFMA / VALU counter passes the
# round-2 verdict asked for (separate runs with --kernel-trace only, as the pool requires).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_fit
mkdir -p $O
for cfg in "gcv 5000 3" "gcvcache 5000 3" "gcvcache 4100 3" "gcv 20000 1"; do
  set -- $cfg
  tag=${1}_n${2}
  rm -rf /tmp/kt_$tag
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$tag -o $tag -- python $R/tools/fit_pmc.py $1 $2 $3 > /tmp/kt_$tag.log 2>&1
  echo "== kernel stats $tag rc=$?"
  find /tmp/kt_$tag -name "*kernel_stats.csv" -exec cp {} $O/fit_${tag}_kernel_stats.csv \;
  head -6 $O/fit_${tag}_kernel_stats.csv | cut -c1-150
done
for set in "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '_')
  for cfg in "gcv 5000" "gcv 20000"; do
    set2=($cfg)
    tag=${set2[0]}_n${set2[1]}
    rm -rf /tmp/pmc_$tag
    timeout 900 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- python $R/tools/fit_pmc.py ${set2[0]} ${set2[1]} 1 > /tmp/pmc_$tag.log 2>&1
    echo "== pmc [$set] $tag rc=$?"; tail -1 /tmp/pmc_$tag.log
    find /tmp/pmc_$tag -name "*counter_collection.csv" -exec cp {} $O/fit_${tag}_pmc_${name}.csv \;
  done
done
python3 - <<'PY'
import csv, collections, glob, json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r03_fit"
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
    for k, d in sorted(ks.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:10]:
        print("   %-48s %s" % (k, {c: "%.3g" % v for c, v in d.items()}))
PY
