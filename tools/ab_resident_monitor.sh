#!/bin/bash
# A/B (upper bound): the on-chip rod kernel with its first sweeps run WITHOUT the convergence monitor (no |update| max, no read of
# the old slot): SRI_DEBUG=resident_skip_c=<coarse sweeps>,resident_skip_f=<fine sweeps>.  NEEDS a throw-away build with that
# switch (EXPERIMENTS R5.14; not in the product).  Same box, alternating processes.
set -e
cd "$(dirname "$0")/.."
out=gpurun_out/r05_ab_resident_monitor.jsonl
: > $out
for cfg in "256 65536" "128 131072" "32 65536" "64 32768"; do
  set -- $cfg
  for rep in 1 2; do
    for v in "" "resident_skip_c=12,resident_skip_f=5" "resident_skip_c=16,resident_skip_f=7"; do
      SRI_DEBUG=$v python3 tools/rod_bench.py --no-cpu --reps 20 --n $1 --rods $2 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); d['variant']='$v' or 'default'; print(json.dumps(d))" >> $out
    done
  done
done
python3 -c "
import json
for l in open('$out'):
    d=json.loads(l); print(d['n'], d['rods'], d['variant'], round(d['rods']/d['value']*1e3,4), 'ms', d['last_update'], d.get('picard_iterations'))
"
