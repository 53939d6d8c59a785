"""Sums a `rocprofv3 --pmc FETCH_SIZE --kernel-trace` pass over tools/r06_forest_big.py: memory-side bytes fetched per dispatch of
the forest kernels, in the order the script launches them (per raster variant: cbs, cbs plain, compact, compact plain; 4
dispatches each).  FETCH_SIZE is in KB and, on gfx950, counts half the bytes of wide streaming reads (MI355X guide, HBM
section): the RATIO between the kernels is what this file is for.
   python tools/r06_forest_pmc_summary.py <counter_collection.csv> <trees> <cells>"""
import csv
import json
import sys

path, trees, cells = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        if r.get("Counter_Name") != "FETCH_SIZE":
            continue
        name = r["Kernel_Name"]
        if "rf_walk_cbs_kernel" in name or "rf_walk_compact_kernel" in name:
            rows.append((int(r["Dispatch_Id"]), "cbs" if "cbs" in name else "compact", float(r["Counter_Value"])))
rows.sort()
# consecutive dispatches of one kernel form a group (a variant of the script: 1 warm-up + 3 timed calls)
groups = []
for _, k, v in rows:
    if groups and groups[-1]["kernel"] == k and len(groups[-1]["values"]) < 4:
        groups[-1]["values"].append(v)
    else:
        groups.append({"kernel": k, "values": [v]})
labels = ["cbs", "cbs plain", "compact", "compact plain"]
rasters = ["8d planes", "8d + 1 % noise", "8d + 10 % noise"]
out = {"source": "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python tools/r06_forest_big.py %d %d" % (trees, int(cells // 20000)),
       "unit": "FETCH_SIZE as reported (KB), mean per dispatch; bytes_per_cell_and_tree = KB x 1024 / cells / trees (uncorrected)", "rasters": {}}
for i, g in enumerate(groups):
    ras = rasters[i // 4] if i // 4 < len(rasters) else "?"
    lab = labels[i % 4]
    mean = sum(g["values"]) / len(g["values"])
    out["rasters"].setdefault(ras, {})[lab] = {"launched_kernel": g["kernel"], "fetch_kb": mean, "dispatches": len(g["values"]),
                                                "bytes_per_cell_and_tree": mean * 1024.0 / cells / trees}
for ras, d in out["rasters"].items():
    if "cbs" in d and "compact" in d and d["cbs"]["fetch_kb"] > 0:
        d["compact_over_cbs"] = d["compact"]["fetch_kb"] / d["cbs"]["fetch_kb"]
print(json.dumps(out, indent=1))
