#!/bin/bash
# round-5 batch 16: EPI_QBC variants (pairs of X in flight, the L2 touch) in the rod solve, N = 256 and 512
L=experimental_gpu_programming_for_a_spectral_numerical_integration_amd/libsri_hip.so
python -m pytest tests/test_rod_gpu.py -m gpu -x -q -k "criterion" > gpurun_out/r05_qbc_tests3.log 2>&1
rc=$?; tail -3 gpurun_out/r05_qbc_tests3.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
for v in default:$L d2t0:build/ab/libsri_d2t0.so d1t1:build/ab/libsri_d1t1.so d3t1:build/ab/libsri_d3t1.so; do
  for n in 256:65536 512:16384; do
    SRI_HIP_LIBRARY=$PWD/${v##*:} python tools/rod_bench.py --n ${n%%:*} --rods ${n##*:} --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${v%%:*}', d['n'], d['ms_total'], d['last_update'])" >> gpurun_out/r05_ab_qbc_variants.log
  done
done
done
SRI_HIP_LIBRARY=$PWD/$L python tools/rod_bench.py --n 256 --rods 65536 --no-cpu --no-fused-criterion 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pass', d['n'], d['ms_total'], d['last_update'])" >> gpurun_out/r05_ab_qbc_variants.log
cat gpurun_out/r05_ab_qbc_variants.log
