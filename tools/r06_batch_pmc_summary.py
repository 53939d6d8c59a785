"""rocprofv3 --pmc passes on tools/r06_batch_probe.py -> per-dispatch counters of tps_small_batch_kernel (sum over its dispatches)."""
import csv, glob, json, os, sys
O = sys.argv[1]
tot, n = {}, 0
for f in glob.glob(os.path.join(O, "batch_fit_pmc_*.csv")):
    for r in csv.DictReader(open(f)):
        if "tps_small_batch_kernel" not in r.get("Kernel_Name", ""):
            continue
        tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
out = {"kernel": "tps_small_batch_kernel", "what": "sums over every dispatch of tools/r06_batch_probe.py (49-fit batches and single fits)", "counters": tot}
c = tot
if c.get("SQ_INSTS_VALU") and c.get("SQ_WAVES"):
    out["valu_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_BUSY_CYCLES"):
    out["valu_busy_share_of_sq_busy"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_BUSY_CYCLES"]
if c.get("SQ_LDS_BANK_CONFLICT") and c.get("SQ_LDS_IDX_ACTIVE"):
    out["lds_bank_conflict_share"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
print(json.dumps(out, indent=1))
