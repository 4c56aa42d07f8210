#!/usr/bin/env python3
"""Condenses the rocprofv3 output of tools/profile_bench.sh into profiles/<tag>_*.{csv,json} (what the judge reads)."""
import csv
import glob
import json
import os
import sys

out, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
os.makedirs(prof, exist_ok=True)
KERNELS = ("apply_kernel", "apply_small_kernel")


def is_apply(name):
    return any(k in name for k in KERNELS)


def find(sub, pat):
    hits = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


summary = {"tag": tag}
line = None
try:
    line = json.loads([ln for ln in open(os.path.join(out, "bench_line.json")) if ln.strip().startswith("{")][-1])
    cfg = line["config"]
    summary["workload"] = cfg["workload"]
    summary["bench_unprofiled"] = {"kernel_ms": line["roofline"]["kernel_ms"], "ms_per_step": line["ms_per_step"],
                                   "value": line["value"], "roofline": {k: line["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac")}}
except Exception as e:  # noqa: BLE001
    summary["bench_line_error"] = str(e)
f = find("trace", "*kernel_stats.csv")
if f:
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(prof, f"{tag}_rocprofv3_kernel_stats.csv"), "w") as o:
        w = csv.writer(o)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([r["Name"][:140], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
    for r in rows:
        if is_apply(r["Name"]):
            summary["kernel"] = r["Name"][:140]
            summary["calls"] = int(r["Calls"])
            summary["avg_ns"] = float(r["AverageNs"])
            summary["min_ns"] = float(r["MinNs"])


def pmc(sub):
    f = find(sub, "*counter_collection.csv")
    acc = {}
    if not f:
        return acc
    for r in csv.DictReader(open(f)):
        if not is_apply(r.get("Kernel_Name", "")):
            continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


fetch, write, sq = pmc("fetch"), pmc("write"), pmc("sq")
if "FETCH_SIZE" in fetch:
    kb = fetch["FETCH_SIZE"][0]
    # FETCH_SIZE is reported in KiB; on gfx950 it counts 128-B streaming requests as 64 B -> x2 (MI355X_MICROARCH.md, HBM)
    summary["FETCH_SIZE_KiB_raw_per_launch"] = kb
    summary["hbm_read_bytes_per_launch_corrected_x2"] = kb * 1024 * 2
if "WRITE_SIZE" in write:
    kb = write["WRITE_SIZE"][0]
    summary["WRITE_SIZE_KiB_raw_per_launch"] = kb
    summary["hbm_write_bytes_per_launch"] = kb * 1024
if "hbm_read_bytes_per_launch_corrected_x2" in summary and "hbm_write_bytes_per_launch" in summary:
    summary["hbm_bytes_per_launch"] = summary["hbm_read_bytes_per_launch_corrected_x2"] + summary["hbm_write_bytes_per_launch"]
for k, (v, n) in sq.items():
    summary[k + "_per_launch"] = v
if "SQ_VALU_MFMA_BUSY_CYCLES" in sq and "GRBM_GUI_ACTIVE" in sq:
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles are summed over all SIMDs (1024)
    gui = sq["GRBM_GUI_ACTIVE"][0] / 8.0
    summary["mfma_busy_frac_of_active_cycles"] = sq["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (gui * 1024.0)
if "hbm_bytes_per_launch" in summary and line is not None:
    import datetime
    cfg = line["config"]
    alg = line["roofline"]["algorithmic_bytes_per_launch"]
    summary["hbm_bytes_over_algorithmic"] = summary["hbm_bytes_per_launch"] / alg
    rows_ = cfg["N"] if cfg["operator"] == "dn" else cfg["N"] - 1
    ld_ = cfg.get("ld", rows_)
    ld_tag = "" if ld_ == rows_ else f"_ld{ld_}"
    rec = {"n": cfg["N"], "rods": cfg["rods_per_gpu"], "operator": cfg["operator"], "ld": ld_,
           "hbm_bytes_per_launch": summary["hbm_bytes_per_launch"], "algorithmic_bytes_per_launch": alg,
           "rocprofv3_kernel_avg_ns": summary.get("avg_ns"), "rocprofv3_kernel_calls": summary.get("calls"),
           "date": datetime.date.today().isoformat(),
           "lib_sources_sha16": line["roofline"].get("lib_sources_sha16"),   # bench.py replays this record only for the same sources
           "source": f"profiles/{tag}_rocprofv3_summary.json (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes)"}
    json.dump(rec, open(os.path.join(prof, f"hbm_traffic_n{cfg['N']}_r{cfg['rods_per_gpu']}_{cfg['operator']}{ld_tag}.json"), "w"), indent=1)
json.dump(summary, open(os.path.join(prof, f"{tag}_rocprofv3_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
