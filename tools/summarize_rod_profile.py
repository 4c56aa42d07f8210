#!/usr/bin/env python3
"""Condenses tools/profile_rod.sh output into profiles/<tag>_rod_rocprofv3_kernel_stats.csv and
profiles/<tag>_fused_step_pmc.json (per-launch means per kernel variant of the integration apply: plain, with the
initial-value term, and the fused Picard step — the last template argument 1)."""
import csv
import glob
import json
import os
import sys

out, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")


def find(sub, pat):
    hits = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


f = find("trace", "*kernel_stats.csv")
if f:
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(prof, f"{tag}_rod_rocprofv3_kernel_stats.csv"), "w") as o:
        w = csv.writer(o)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([r["Name"][:150], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])


def pmc(sub):
    f = find(sub, "*counter_collection.csv")
    acc = {}
    if not f:
        return acc
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if "apply_kernel" not in name:
            continue
        acc.setdefault(name[:150], {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: {c: (sum(v) / len(v)) for c, v in d.items()} | {"launches": max(len(v) for v in d.values())} for k, d in acc.items()}


res = {"source": "tools/profile_rod.sh: rocprofv3 --kernel-trace --pmc <one group per pass> -- python tools/step_bench.py --n 256 "
                 "--rods 65536 (255-row integration operator, per-launch means). Kernel template arguments: <Geo, DMA, RANK1, TAIL, "
                 "STAMP, ABL, EPI, ASC, LEAN>; EPI = 1 is the fused Picard step, LEAN the twin without edge handling. FETCH_SIZE is in KiB and counts 128-B requests as 64 B on "
                 "gfx950 (x2, MI355X_MICROARCH.md HBM); WRITE_SIZE in KiB.",
       "kernels": {}}
for sub in ("lds", "fetch", "write"):
    for k, d in pmc(sub).items():
        res["kernels"].setdefault(k, {}).update({c: v for c, v in d.items() if c != "launches"})
        res["kernels"][k]["launches_" + sub] = d["launches"]
f = find("steptrace", "*kernel_stats.csv")
if f:
    for r in csv.DictReader(open(f)):
        k = r["Name"][:150]
        if k in res["kernels"]:
            res["kernels"][k]["avg_ns"] = float(r["AverageNs"])
            res["kernels"][k]["calls"] = int(r["Calls"])
m, rods = 255, 65536
alg = {"plain / integrate": 64.0 * m * rods, "fused step": 64.0 * m * rods + 72.0 * rods}
for k, d in res["kernels"].items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_read_bytes_corrected_x2"] = d["FETCH_SIZE"] * 1024 * 2
        d["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024
        d["hbm_bytes"] = d["hbm_read_bytes_corrected_x2"] + d["hbm_write_bytes"]
        targs = k[k.rfind(">,") + 2:k.rfind(">(")].split(",")          # ... DMA, RANK1, TAIL, STAMP, ABL, EPI, ASC, LEAN
        fused = len(targs) >= 3 and targs[-3].strip() == "1"
        d["fused_picard_step"] = fused
        d["algorithmic_bytes"] = alg["fused step"] if fused else alg["plain / integrate"]
        d["hbm_over_algorithmic"] = d["hbm_bytes"] / d["algorithmic_bytes"]
    if "avg_ns" in d:
        d["TFLOPs"] = 8.0 * m * m * rods / d["avg_ns"] / 1e3
json.dump(res, open(os.path.join(prof, f"{tag}_fused_step_pmc.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
