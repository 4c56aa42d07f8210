#!/bin/bash
# ab_fused_criterion_variants.sh — EPI_QBC builds with one / two / three row pairs of X in flight in the rod solve, N = 256 and 512.
# Variant libraries first (on the build host):  make -C <pkg>/csrc variant NAME=d1 DEFS=-DSRI_QBC_DEPTH=1  (-> build/libsri_hip_d1.so), same for d3.
# (The round-5 run also had a build with an L2 prefetch of the tile's X lines — "t1" in profiles/r05_ab_fused_criterion_variants.log;
#  that code was removed after this measurement.)
L=experimental_gpu_programming_for_a_spectral_numerical_integration_amd/libsri_hip.so
python -m pytest tests/test_rod_gpu.py -m gpu -x -q -k "criterion" > gpurun_out/r05_rod_tests_criterion.log 2>&1
rc=$?; tail -3 gpurun_out/r05_rod_tests_criterion.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
for v in default:$L d1:build/libsri_hip_d1.so d3:build/libsri_hip_d3.so; do
  for n in 256:65536 512:16384; do
    SRI_HIP_LIBRARY=$PWD/${v##*:} python tools/rod_bench.py --n ${n%%:*} --rods ${n##*:} --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${v%%:*}', d['n'], d['ms_total'], d['last_update'])" >> gpurun_out/r05_ab_qbc_variants.log
  done
done
done
SRI_HIP_LIBRARY=$PWD/$L python tools/rod_bench.py --n 256 --rods 65536 --no-cpu --no-fused-criterion 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pass', d['n'], d['ms_total'], d['last_update'])" >> gpurun_out/r05_ab_qbc_variants.log
cat gpurun_out/r05_ab_qbc_variants.log
