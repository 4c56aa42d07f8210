#!/usr/bin/env python3
"""Condenses tools/profile_rod_pmc.sh's rocprofv3 output into profiles/<tag>_rod_pmc_by_kernel.json: per kernel of the rod
solve the mean duration, MFMA-busy share of the active cycles, wait share of the wave cycles, LDS conflicts and HBM bytes
(FETCH_SIZE in KiB, x2 on gfx950 as MI355X_MICROARCH.md prescribes, + WRITE_SIZE in KiB)."""
import csv
import glob
import json
import os
import re
import sys

out, tag, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(sub, pat):
    hits = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


def short(name):
    name = re.sub(r"^void ", "", name)
    return name[:110]


res = {"tag": tag, "command": "python " + cmd, "kernels": {}}
f = find("trace", "*kernel_stats.csv")
if f:
    for r in csv.DictReader(open(f)):
        res["kernels"][short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "share_of_gpu_time": float(r["Percentage"]) / 100.0}


def pmc(sub):
    f = find(sub, "*counter_collection.csv")
    acc = {}
    if not f:
        return acc
    for r in csv.DictReader(open(f)):
        acc.setdefault(short(r.get("Kernel_Name", "")), {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


sq, fetch, write = pmc("sq"), pmc("fetch"), pmc("write")
for k, rec in res["kernels"].items():
    s = sq.get(k, {})
    if "SQ_VALU_MFMA_BUSY_CYCLES" in s and s.get("GRBM_GUI_ACTIVE"):
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over all 1024 SIMDs
        rec["mfma_busy_frac_of_active_cycles"] = s["SQ_VALU_MFMA_BUSY_CYCLES"] / (s["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    if s.get("SQ_WAVE_CYCLES"):
        rec["wait_any_frac_of_wave_cycles"] = s.get("SQ_WAIT_ANY", 0.0) / s["SQ_WAVE_CYCLES"]
        rec["wait_inst_any_frac_of_wave_cycles"] = s.get("SQ_WAIT_INST_ANY", 0.0) / s["SQ_WAVE_CYCLES"]
    if "SQ_LDS_BANK_CONFLICT" in s:
        rec["lds_bank_conflict_cycles"] = s["SQ_LDS_BANK_CONFLICT"]
        rec["lds_idx_active_cycles"] = s.get("SQ_LDS_IDX_ACTIVE")
    if "FETCH_SIZE" in fetch.get(k, {}):
        rec["hbm_read_bytes_x2"] = fetch[k]["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in write.get(k, {}):
        rec["hbm_write_bytes"] = write[k]["WRITE_SIZE"] * 1024
    if "hbm_read_bytes_x2" in rec and "hbm_write_bytes" in rec:
        rec["hbm_TBps_at_avg_duration"] = (rec["hbm_read_bytes_x2"] + rec["hbm_write_bytes"]) / (rec["avg_us"] * 1e-6) / 1e12
res["kernels"] = dict(sorted(res["kernels"].items(), key=lambda kv: -kv[1]["share_of_gpu_time"]))
res["kernels"] = {k: v for k, v in list(res["kernels"].items())[:12]}
path = os.path.join(root, "profiles", f"{tag}_rod_pmc_by_kernel.json")
json.dump(res, open(path, "w"), indent=1)
print(json.dumps(res, indent=1))
