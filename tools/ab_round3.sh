#!/bin/bash
# A/B of the round-3 variants (run on the GPU box): tools/ab_round3.sh
PKG=experimental_gpu_programming_for_a_spectral_numerical_integration_amd
LIBS="base=build/libsri_hip_base.so new=$PKG/libsri_hip.so tall4=build/libsri_hip_tall4.so tall4p=build/libsri_hip_tall4p.so"
for cfg in "--n 256 --rods 65536 --operator 0" "--n 256 --rods 65536 --operator 1" "--n 1024 --rods 8192 --operator 0"; do
  python tools/ab_apply.py $cfg --rounds 5 --reps 200 --libs $LIBS
done
