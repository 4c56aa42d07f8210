#!/bin/bash
# A/B of the round-3 kernel against the round-2 one and against itself with the split stage wait (run on the GPU box)
PKG=experimental_gpu_programming_for_a_spectral_numerical_integration_amd
LIBS="round2=build/libsri_hip_base.so round3=$PKG/libsri_hip.so"
for cfg in "--n 256 --rods 65536 --operator 0" "--n 256 --rods 65536 --operator 1" "--n 256 --rods 65536 --operator 1 --ld 256" "--n 1024 --rods 8192 --operator 0" "--n 1024 --rods 8192 --operator 1" "--n 128 --rods 65536 --operator 0" "--n 64 --rods 262144 --operator 0"; do
  python tools/ab_apply.py $cfg --rounds 5 --reps 200 --libs $LIBS
done
