#!/bin/bash
# ab_mini_strip_64_rows.sh — the 64-row geometry with the mini strip (65 ... 68 rows): the geometry tests, then A/B against the 96-row block
# (SRI_DEBUG=mini=0) and a build with three ring slots:  make -C <pkg>/csrc variant NAME=fm3 DEFS=-DSRI_FLAT_MINI_STAGES=3
L=experimental_gpu_programming_for_a_spectral_numerical_integration_amd/libsri_hip.so
python -m pytest tests/test_apply_gpu.py -m gpu -x -q -k "geometry_above or parts_give" > gpurun_out/r05_flatmini_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r05_flatmini_tests.log
[ $rc -eq 0 ] || exit $rc
for n in 65 66 68; do
  python tools/ab_apply.py --n $n --rods 262144 --rounds 5 --reps 100 --libs "mini2=$L" "g96=$L,SRI_DEBUG=mini=0" "mini3=build/libsri_hip_fm3.so" >> gpurun_out/r05_ab_flatmini.log 2>&1 || exit 1
done
python tools/ab_apply.py --n 65 --rods 16384 --rounds 5 --reps 200 --libs "mini2=$L" "g96=$L,SRI_DEBUG=mini=0" "mini3=build/libsri_hip_fm3.so" >> gpurun_out/r05_ab_flatmini.log 2>&1 || exit 1
python tools/ab_apply.py --n 67 --rods 262144 --operator 1 --rounds 5 --reps 100 --libs "mini2=$L" "g96=$L,SRI_DEBUG=mini=0" "mini3=build/libsri_hip_fm3.so" >> gpurun_out/r05_ab_flatmini.log 2>&1 || exit 1
python tools/ab_apply.py --n 321 --rods 40000 --rounds 5 --reps 60 --libs "mini2=$L" "g96=$L,SRI_DEBUG=mini=0" >> gpurun_out/r05_ab_flatmini.log 2>&1 || exit 1
grep -v amdgpu.ids gpurun_out/r05_ab_flatmini.log
