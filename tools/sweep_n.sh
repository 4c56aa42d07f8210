#!/bin/bash
# sweep_n.sh <tag> — the apply at grid sizes between the powers of two, ~1 GiB panels each (run on the GPU box):
#   profiles/<tag>_sweep_n.jsonl, one bench.py line per N (roofline.frac = fraction of the bound; config.kernel = what ran)
tag=${1:-r04}
out=gpurun_out/${tag}_sweep_n.jsonl
: > $out
for n in ${SWEEP_NS:-32 48 64 96 128 160 192 256 257 300 320 384 512 513 576 640 768 896 1024 1025 1088}; do
  rods=$(python3 -c "print(max(64, (2**29 // ($n * 8 * 4)) // 32 * 32))")
  timeout -k 10 120 python bench.py --n $n --rods $rods --steps 300 --warmup 30 --ramp-ms 600 --cpu-seconds 0 --no-verify --long-run 0 >> $out 2>> gpurun_out/${tag}_sweep_n.err || echo "{\"n\": $n, \"failed\": true}" >> $out
done
python3 - <<PY
import json
for l in open("$out"):
    d = json.loads(l)
    if d.get("failed"): print(d); continue
    c, r = d["config"], d["roofline"]
    print(c["N"], c["rods_per_gpu"], round(r["frac"], 3), r["bound"], c["kernel"], "parity", d["parity"]["ok"])
PY
