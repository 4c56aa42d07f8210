#!/bin/bash
# A/B of the odd-row-count (TAIL) variants against a baseline build in ONE process (run on the GPU box):
#   build/libsri_hip_base.so = the library of the commit to compare with (git worktree + make, see tools/ab_round3.sh)
PKG=experimental_gpu_programming_for_a_spectral_numerical_integration_amd
LIBS="base=build/libsri_hip_base.so new=$PKG/libsri_hip.so"
for cfg in "--n 256 --rods 65536 --operator 1" "--n 256 --rods 65536 --operator 1 --ld 256" "--n 256 --rods 65536 --operator 0" "--n 1024 --rods 8192 --operator 1" "--n 1024 --rods 8192 --operator 1 --ld 1024" "--n 128 --rods 65536 --operator 1" "--n 64 --rods 262144 --operator 1" "--n 512 --rods 16384 --operator 2"; do
  python tools/ab_apply.py $cfg --rounds 5 --reps 200 --libs $LIBS || exit 1
done
