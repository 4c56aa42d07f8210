#!/bin/bash
# round 5: the 128-row and 96-row blocks as ONE eight-wave workgroup per CU (three ring slots, the tall geometry's stream) against the
# shipped four-wave / two-workgroup geometries — build/libsri_hip_midalt.so = make -C csrc variant NAME=midalt DEFS="-DSRI_GEO_MID_ALT -DSRI_GEO_96_ALT"
PKG=experimental_gpu_programming_for_a_spectral_numerical_integration_amd
for cfg in "--n 128 --rods 131072" "--n 128 --rods 131072 --operator 1" "--n 112 --rods 149760" "--n 96 --rods 174752" "--n 80 --rods 209664" "--n 96 --rods 174752 --operator 1"; do
  python tools/ab_apply.py $cfg --rounds 5 --reps 200 --libs own=$PKG/libsri_hip.so,opt:1=0 alt=build/libsri_hip_midalt.so,opt:1=0 || exit 1
done
