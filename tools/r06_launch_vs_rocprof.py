"""The bench line's HIP-event launch times beside rocprofv3's per-kernel averages of the SAME run (round-5 verdict item 6):
python tools/r06_launch_vs_rocprof.py <bench line json> <kernel_stats.csv>"""
import csv, json, sys
line = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
stats = {r["Name"]: r for r in csv.DictReader(open(sys.argv[2]))}
out = {"source": "one run of `rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline` on one box",
       "ms_per_step": line.get("ms_per_step"), "kernels": []}
for row in line.get("kernels", []):
    key = row["kernel"].split(" ")[0].split("<")[0]
    hits = [(n, r) for n, r in stats.items() if key in n]
    if not hits:
        continue
    n, r = max(hits, key=lambda q: float(q[1]["TotalDurationNs"]))
    avg_ms = float(r["AverageNs"]) / 1e6
    e = {"kernel": row["kernel"], "launch_ms_hip_events": row["launch_ms"], "rocprofv3_kernel": n[:80], "rocprofv3_calls": int(r["Calls"]),
         "rocprofv3_avg_ms": avg_ms, "rocprofv3_min_ms": float(r["MinNs"]) / 1e6, "rocprofv3_max_ms": float(r["MaxNs"]) / 1e6}
    if row.get("frac") and row["launch_ms"]:
        e["frac_from_hip_events"] = row["frac"]
        e["frac_from_rocprofv3_avg"] = row["frac"] * row["launch_ms"] / avg_ms
    out["kernels"].append(e)
print(json.dumps(out, indent=1))
