#!/bin/bash
# A/B of the 32-node on-chip rod kernel's wave geometry (SRI_DEBUG=resident_tn=...): 16 rods per wave x 4 waves (default),
# 8 rods per wave x 8 waves (16 waves per CU), 8 rods per wave x 4 waves (3 workgroups per CU).  Same box, alternating processes.
# NEEDS a throw-away build with the `resident_tn` switch in resident_integrate's `case 2` (EXPERIMENTS R5.13: <2,2,8> / <2,2,4>
# instantiations; not in the product).
set -e
cd "$(dirname "$0")/.."
out=gpurun_out/r05_ab_resident_waves.jsonl
: > $out
for cfg in "256 65536" "128 131072" "32 65536" "24 65536"; do
  set -- $cfg
  for rep in 1 2; do
    for v in "" "resident_tn=2" "resident_tn=2x4"; do
      SRI_DEBUG=$v python3 tools/rod_bench.py --no-cpu --reps 20 --n $1 --rods $2 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); d['variant']='$v' or 'default'; print(json.dumps(d))" >> $out
    done
  done
done
python3 -c "
import json
for l in open('$out'):
    d=json.loads(l); print(d['n'], d['rods'], d['variant'], round(d['rods']/d['value']*1e3,4), 'ms', d['last_update'])
"
