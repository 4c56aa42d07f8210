#!/bin/bash
# ab_plan_calibration.sh — does the first-call verdict of the column plan survive a cold device?  Six fresh processes per library and
# shape (every process calibrates anew): what ran and at which fraction.  build/libsri_hip_mincal.so = the library of commit 10f56df
# (verdict from the minima of own-plan-plan-own, no warm-up pair).
for lib in experimental_gpu_programming_for_a_spectral_numerical_integration_amd/libsri_hip.so build/libsri_hip_mincal.so; do
  for spec in "320 52416" "385 43552" "256 66560"; do
    set -- $spec
    for i in 1 2 3 4 5 6; do
      SRI_HIP_LIBRARY=$PWD/$lib python bench.py --n $1 --rods $2 --steps 300 --warmup 30 --ramp-ms 600 --cpu-seconds 0 --no-verify --long-run 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$lib'.split('/')[-1], d['config']['N'], d['config']['rods_per_gpu'], round(d['roofline']['frac'],3), d['config']['kernel'].replace('sri_apply_mfma_f64',''))"
    done
  done
done
