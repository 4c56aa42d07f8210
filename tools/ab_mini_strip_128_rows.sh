#!/bin/bash
# ab_mini_strip_128_rows.sh — the 128-row geometry with the mini strip (129 ... 132 rows): the geometry tests, then A/B against the 160 x 64 block (SRI_DEBUG=mini=0)
L=experimental_gpu_programming_for_a_spectral_numerical_integration_amd/libsri_hip.so
python -m pytest tests/test_apply_gpu.py -m gpu -x -q -k "geometry_above or parts_give" > gpurun_out/r05_midmini_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r05_midmini_tests.log
[ $rc -eq 0 ] || exit $rc
for n in 129 130 132; do
  python tools/ab_apply.py --n $n --rods 65536 --rounds 5 --reps 100 --libs "mini=$L" "g160=$L,SRI_DEBUG=mini=0" >> gpurun_out/r05_ab_midmini.log 2>&1 || exit 1
done
python tools/ab_apply.py --n 129 --rods 65536 --operator 1 --rounds 5 --reps 100 --libs "mini=$L" "g160=$L,SRI_DEBUG=mini=0" >> gpurun_out/r05_ab_midmini.log 2>&1 || exit 1
python tools/ab_apply.py --n 385 --rods 32768 --rounds 5 --reps 60 --libs "mini=$L" "g160=$L,SRI_DEBUG=mini=0" >> gpurun_out/r05_ab_midmini.log 2>&1 || exit 1
python tools/ab_apply.py --n 897 --rods 8192 --rounds 5 --reps 60 --libs "mini=$L" "g160=$L,SRI_DEBUG=mini=0" >> gpurun_out/r05_ab_midmini.log 2>&1 || exit 1
cat gpurun_out/r05_ab_midmini.log
