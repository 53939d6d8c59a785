"""profiles/r03_members_pmc_summary.json (tools/r03_members_pmc.sh: counters summed over the 2 dispatches of each member
kernel on a side x side grid) -> profiles/r03_members_pmc_derived.json: per-unit instruction counts and pipe utilisations.
    python tools/r03_pmc_derive.py      (reads profiles/r03_members_pmc_summary.json and r03_members_pmc_units.json)
Unit conventions (rocprofiler-sdk counter_defs.yaml): SQ_INSTS_* count wave-instructions; a wave64 VALU instruction
occupies its SIMD for 4 cycles (16 lanes / clk); GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_LDS_IDX_ACTIVE counts
LDS-array cycles summed over the CUs; FETCH_SIZE / WRITE_SIZE are KB (FETCH_SIZE doubled for gfx950, MI355X guide)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
U = json.load(open(os.path.join(ROOT, "profiles", "r03_members_pmc_units.json")))      # written by tools/r03_members_pmc.py
side = int(U["side"])
units = {"gbm": U.get("gbm"), "rf": U.get("rf"), "svr": U.get("svr")}      # trees; sum of tree depths (levels per cell); support vectors
S = json.load(open(os.path.join(ROOT, "profiles", "r03_members_pmc_summary.json")))
# per member the kernel that did the work (gbm launches a probe and two kernels of which one returns at once)
best = {}
for k, d in S.items():
    kind = "gbm" if "gbm" in k else "rf" if "rf_" in k else "svr"
    t = d["GRBM_GUI_ACTIVE"]["sum"] / d["GRBM_GUI_ACTIVE"]["dispatches"]
    if kind not in best or t > best[kind][1]: best[kind] = (k, t)
S = {k: d for k, d in S.items() if any(k == b[0] for b in best.values())}
cells = float(side) * side
out = {"grid": [side, side], "note": __doc__.split("\n")[0]}
for k, d in S.items():
    v = {c: x["sum"] / x["dispatches"] for c, x in d.items()}              # per dispatch
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0                                         # elapsed cycles of one dispatch
    kind = "gbm" if "gbm" in k else "rf" if "rf_" in k else "svr"
    r = {"elapsed_cycles": cyc,
         "valu_issue_utilisation": v["SQ_INSTS_VALU"] * 4.0 / (1024.0 * cyc),
         "salu_per_valu": v["SQ_INSTS_SALU"] / v["SQ_INSTS_VALU"],
         "lds_array_busy": v["SQ_LDS_IDX_ACTIVE"] / (256.0 * cyc),
         "lds_bank_conflict_share_of_lds_cycles": v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"],
         "lds_cycles_per_lds_instruction": v["SQ_LDS_IDX_ACTIVE"] / v["SQ_INSTS_LDS"],
         "lds_cmd_fifo_full_share": v["SQ_LDS_CMD_FIFO_FULL"] / (256.0 * cyc),
         "hbm_bytes_per_cell_fetch_x2_plus_write": (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0 / cells,
         "valu_wave_instructions": v["SQ_INSTS_VALU"], "lds_wave_instructions": v["SQ_INSTS_LDS"], "salu_wave_instructions": v["SQ_INSTS_SALU"]}
    if units[kind]:
        per = cells * units[kind] / 64.0                                     # wave-level (cell, unit) pairs
        r["valu_per_cell_unit"] = v["SQ_INSTS_VALU"] / per
        r["lds_per_cell_unit"] = v["SQ_INSTS_LDS"] / per
        r["salu_per_cell_unit"] = v["SQ_INSTS_SALU"] / per
        r["unit"] = {"gbm": "tree", "rf": "tree level (full depth)", "svr": "support vector"}[kind]
    if kind == "rf":      # two LDS instructions per walk and level walked (node record, key): the levels the waves really descended
        r["levels_walked_per_cell"] = v["SQ_INSTS_LDS"] * 64.0 / cells / 2.0
        r["levels_full_depth_per_cell"] = units["rf"]
    out[k] = r
json.dump(out, open(os.path.join(ROOT, "profiles", "r03_members_pmc_derived.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
