# Round 3: PMC passes on the heavy member kernels (one counter set per run, --kernel-trace only), summed per kernel.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3/pmc
mkdir -p $O
SIDE=${1:-10000}
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_CYCLES_VALU"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/mp_$i -o p -- python $R/tools/r03_members_pmc.py $SIDE ${2:-brv} > /tmp/mp_$i.log 2>&1
  echo "== $C rc=$?"; tail -2 /tmp/mp_$i.log
  cp /tmp/mp_$i/p_counter_collection.csv $O/pass${i}_counter_collection.csv 2>/dev/null
done
python3 - <<'PY'
import csv, collections, glob, os, json
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r3/pmc"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in sorted(glob.glob(O + "/pass*_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].split("::")[-1].replace("void ", "")
        if not any(s in k for s in ("gbm_lutreg", "gbm_coherent", "rf_walk", "svr_kernel", "svr_rt_kernel")): continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k][row["Counter_Name"]] += 1
res = {k: {c: {"sum": v, "dispatches": cnt[k][c]} for c, v in d.items()} for k, d in agg.items()}
json.dump(res, open(O + "/summary.json", "w"), indent=1)
for k, d in sorted(res.items()):
    print(k)
    for c, v in sorted(d.items()): print("   %-28s %.5g  (%d dispatches)" % (c, v["sum"], v["dispatches"]))
PY
