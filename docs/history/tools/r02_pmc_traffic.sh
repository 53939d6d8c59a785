#!/bin/bash
# round 2: HBM bytes per cell of the ensemble / TPS-evaluation kernels from the PMC counters, as the guide prescribes
# (separate --pmc passes with --kernel-trace only; FETCH_SIZE doubled on gfx950).  Run on the GPU box through gpurun.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_final; mkdir -p $O
P=/tmp/pmc_probe
/opt/rocm/bin/hipcc -O2 -I$R/include -o $P $R/tools/pmc_probe.cpp -L$R/machisplin_amd -lmachisplin_hip -Wl,-rpath,$R/machisplin_amd || exit 1
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -o p -- $P 4000 > /tmp/pmc_$C.log 2>&1
  echo "== $C rc=$?"
  find /tmp/pmc_$C -name "*counter_collection.csv" -exec cp {} $O/pmc_probe_${C}_counter_collection.csv \;
done
python3 - <<'PY'
import csv, collections, json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r02_final"
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f"{O}/pmc_probe_{c}_counter_collection.csv")):
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
    print(k[:70], {a: round(b, 3) for a, b in res["kernels"][k].items()})
json.dump(res, open(O + "/pmc_traffic.json", "w"), indent=1)
PY
