"""The bench line's HIP-event launch times beside rocprofv3's durations of the SAME run (round-5 verdict item 6):
python tools/r06_launch_vs_rocprof.py <bench line json> <kernel_stats.csv> [<kernel_trace.csv>]
With the trace, a kernel's full-grid dispatches are the ones that last at least half as long as its longest (the station-residual
calls and the fit's small launches of the same kernels are left out) and their mean is what the line's launch_ms is held against."""
import csv, json, sys
line = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
stats = {r["Name"]: r for r in csv.DictReader(open(sys.argv[2]))}
trace = {}
if len(sys.argv) > 3:
    for r in csv.DictReader(open(sys.argv[3])):
        trace.setdefault(r["Kernel_Name"], []).append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6)
out = {"source": "one run of `MHS_RESERVE_KEEP=1 MHS_BENCH_SKIP_F64=1 MHS_BENCH_SKIP_SENSITIVITY=1 MHS_BENCH_SKIP_FITTED=1 rocprofv3 --kernel-trace --stats -- python bench.py "
                 "--steps 5 --warmup 1 --no-cpu-baseline` on one box: the timed steps (+ warm-up and the reservation calibration's steps) only",
       "ms_per_step": line.get("ms_per_step"), "value": line.get("value"), "kernels": []}
for row in line.get("kernels", []):
    key = row["kernel"].split(" ")[0].split("<")[0]
    hits = [(n, r) for n, r in stats.items() if key in n]
    if not hits:
        continue
    n, r = max(hits, key=lambda q: float(q[1]["TotalDurationNs"]))
    e = {"kernel": row["kernel"], "launch_ms_hip_events": row["launch_ms"], "rocprofv3_kernel": n[:80], "rocprofv3_calls": int(r["Calls"]),
         "rocprofv3_avg_ms_all_dispatches": float(r["AverageNs"]) / 1e6, "rocprofv3_max_ms": float(r["MaxNs"]) / 1e6}
    ref = e["rocprofv3_avg_ms_all_dispatches"]
    d = trace.get(n)
    if d:
        full = [x for x in d if x >= 0.5 * max(d)]
        e["rocprofv3_full_grid_dispatches"] = len(full)
        e["rocprofv3_full_grid_mean_ms"] = sum(full) / len(full)
        e["rocprofv3_full_grid_min_ms"], e["rocprofv3_full_grid_max_ms"] = min(full), max(full)
        ref = e["rocprofv3_full_grid_mean_ms"]
    if row.get("frac") and row["launch_ms"]:
        e["frac_from_hip_events"] = row["frac"]
        e["frac_from_rocprofv3"] = row["frac"] * row["launch_ms"] / ref
    out["kernels"].append(e)
print(json.dumps(out, indent=1))
