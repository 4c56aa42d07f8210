#!/bin/bash
# A/B of the round-3 kernel against the round-2 one in ONE process, seven shapes (run on the GPU box).
#   round2 = libsri_hip.so of round 2's HEAD:  git worktree add /tmp/r02 84068c1 && make -C /tmp/r02/<pkg>/csrc &&
#            cp /tmp/r02/<pkg>/libsri_hip.so build/libsri_hip_round2.so
PKG=experimental_gpu_programming_for_a_spectral_numerical_integration_amd
LIBS="round2=build/libsri_hip_round2.so round3=$PKG/libsri_hip.so"
for cfg in "--n 256 --rods 65536 --operator 0" "--n 256 --rods 65536 --operator 1" "--n 256 --rods 65536 --operator 1 --ld 256" "--n 1024 --rods 8192 --operator 0" "--n 1024 --rods 8192 --operator 1" "--n 128 --rods 65536 --operator 0" "--n 64 --rods 262144 --operator 0"; do
  python tools/ab_apply.py $cfg --rounds 5 --reps 200 --libs $LIBS
done
