#!/bin/bash
# ab_plan_calibration.sh — does the verdict of the column plan survive a cold device?  Fresh processes (every process calibrates anew)
# of the sweep's shapes: what ran in the timed region and at which fraction.
for spec in "320 52416" "385 43552" "256 66560" "321 52256"; do
  set -- $spec
  for i in 1 2 3 4 5 6; do
    python bench.py --n $1 --rods $2 --steps 300 --warmup 30 --ramp-ms 600 --cpu-seconds 0 --no-verify --long-run 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['config']['N'], d['config']['rods_per_gpu'], round(d['roofline']['frac'],3), d['config']['kernel'].replace('sri_apply_mfma_f64',''))"
  done
done
