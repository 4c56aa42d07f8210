#!/usr/bin/env python3
"""rod_timeline.py <rocprofv3 kernel_trace.csv> — the kernels of the LAST rod solve in a trace of tools/rod_bench.py, in launch order:
start offset, duration and the idle gap in front of each (where a solve's wall time goes besides its kernels).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/rodtl -- python tools/rod_bench.py --n 256 --rods 65536 --reps 3 --no-cpu
    python tools/rod_timeline.py gpurun_out/rodtl/*/*_kernel_trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a solve starts with the on-chip coarse kernel (rod_resident_kernel); rod_bench's breakdown section follows the timed solves
starts = [i for i, r in enumerate(rows) if "rod_resident_kernel" in r["Kernel_Name"]]
if len(starts) < 2:
    print("no two solves in the trace")
    sys.exit(1)
# the last COMPLETE solve: from its coarse kernel to the kernel in front of the next one (walk back over the fill kernel that precedes it)
a, b = starts[-2], starts[-1]
while a > 0 and "fill_columns" in rows[a - 1]["Kernel_Name"]:
    a -= 1
while b > 0 and "fill_columns" in rows[b - 1]["Kernel_Name"]:
    b -= 1
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"]
    name = name[:name.find("(")] if "(" in name else name
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:7.1f} us  {name[-110:]}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"kernels {busy / 1e3:.1f} us busy in {(prev_end - t0) / 1e3:.1f} us from the first kernel's start to the last one's end")
