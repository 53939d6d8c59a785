"""rocprofv3 --pmc passes on one GCV fit -> per-kernel sums of the MFMA counters: python tools/r06_fit_pmc_summary.py n file.csv ..."""
import csv, json, sys
n = int(sys.argv[1])
acc = {}
for f in sys.argv[2:]:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc.setdefault(k, {}).setdefault(r["Counter_Name"], 0.0)
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
rows = []
for k, c in acc.items():
    e = {"kernel": k, **c}
    if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and c.get("SQ_BUSY_CYCLES"):
        e["mfma_busy_share_of_sq_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CYCLES"]
    if c.get("SQ_INSTS_VALU_MFMA_MOPS_F64"):
        e["mfma_f64_flop"] = c["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512.0      # one MOPS unit = 512 flop (rocprofiler-sdk counter_defs)
    rows.append(e)
rows.sort(key=lambda e: -e.get("SQ_BUSY_CYCLES", 0.0))
tot = sum(e.get("mfma_f64_flop", 0.0) for e in rows)
print(json.dumps({"n": n, "what": "one GCV fit (mhs_tps_fit, 32-column route) under rocprofv3 --pmc, counters summed over each kernel's dispatches",
                  "mfma_f64_flop_total": tot, "model_flop_4_3_m3": 4.0 / 3.0 * (n - 3) ** 3, "kernels": rows[:12]}, indent=1))
