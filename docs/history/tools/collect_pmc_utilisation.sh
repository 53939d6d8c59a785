cd /tmp && export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT/tools/pmc_probe
[ -x $P ] || /opt/rocm/bin/hipcc -O2 -I$GRAFT_REPO_ROOT/include -o $P $GRAFT_REPO_ROOT/tools/pmc_probe.cpp -L$GRAFT_REPO_ROOT/machisplin_amd -lmachisplin_hip -Wl,-rpath,$GRAFT_REPO_ROOT/machisplin_amd
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof
for C in LdsUtil LdsBankConflict "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  T=$(echo $C | tr ' ' '_')
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$T -o p -- $P 4000 > /tmp/pmc_$T.log 2>&1
  echo "== $C rc=$?"
  cp /tmp/pmc_$T/p_counter_collection.csv $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_probe_${T}_counter_collection.csv 2>/dev/null
done
python3 - <<'PY'
import csv, collections, glob, os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof/pmc_probe_*_counter_collection.csv")):
    if "FETCH" in f or "WRITE" in f: continue
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        agg[(row["Kernel_Name"].split("(")[0][-40:], row["Counter_Name"])].append(float(row["Counter_Value"]))
    print(os.path.basename(f))
    for (k, c), v in sorted(agg.items()):
        if any(s in k for s in ("gbm_lutreg", "rf_walk", "svr_kernel", "tps_ff", "lm_kernel")):
            print("   %-42s %-22s %.4g" % (k, c, sum(v) / len(v)))
PY
