cd /tmp && export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT/tools/pmc_probe
[ -x $P ] || /opt/rocm/bin/hipcc -O2 -I$GRAFT_REPO_ROOT/include -o $P $GRAFT_REPO_ROOT/tools/pmc_probe.cpp -L$GRAFT_REPO_ROOT/machisplin_amd -lmachisplin_hip -Wl,-rpath,$GRAFT_REPO_ROOT/machisplin_amd
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -o p -- $P 4000 > /tmp/pmc_$C.log 2>&1
  echo "== $C rc=$?"
  cp /tmp/pmc_$C/p_counter_collection.csv $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_probe_${C}_counter_collection.csv
done
python3 - <<'PY'
import csv, collections, json, os
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f"/tmp/pmc_{c}/p_counter_collection.csv")):
        agg[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    for k, v in agg.items():
        out.setdefault(k, {})[c + "_KB_per_dispatch"] = v
cells = 4000 * 4000
res = {"probe": "tools/pmc_probe 4000 (16e6 cells, 5000 knots, 10000 gbm trees, 3000 SVs, 50 rf trees, 3 float32 planes)", "cells": cells, "kernels": {}}
for k, d in out.items():
    f = sum(d.get("FETCH_SIZE_KB_per_dispatch", [0])) / max(1, len(d.get("FETCH_SIZE_KB_per_dispatch", [0])))
    w = sum(d.get("WRITE_SIZE_KB_per_dispatch", [0])) / max(1, len(d.get("WRITE_SIZE_KB_per_dispatch", [0])))
    res["kernels"][k] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
                         "fetch_bytes_per_cell_x2_corrected": 2 * f * 1024 / cells, "write_bytes_per_cell": w * 1024 / cells}
    print(k[:60], res["kernels"][k])
json.dump(res, open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof/pmc_traffic.json", "w"), indent=1)
PY
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o cfg3 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof/bench_under_rocprof.json 2>/tmp/kst.log )
echo "rocprof bench rc=$?"
cp /tmp/kst/cfg3_kernel_stats.csv gpurun_out/prof/cfg3_rocprofv3_kernel_stats.csv
cp /tmp/kst/cfg3_domain_stats.csv gpurun_out/prof/cfg3_rocprofv3_domain_stats.csv
head -12 gpurun_out/prof/cfg3_rocprofv3_kernel_stats.csv | cut -c1-160
timeout 900 python bench.py > gpurun_out/prof/bench_cfg3_n1.json 2>/dev/null
tail -c 400 gpurun_out/prof/bench_cfg3_n1.json
