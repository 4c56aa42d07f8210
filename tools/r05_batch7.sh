#!/bin/bash
PKG=experimental_gpu_programming_for_a_spectral_numerical_integration_amd
python -m pytest tests/test_apply_gpu.py tests/test_rod_gpu.py tests/test_host_panels_gpu.py -m gpu -x -q > gpurun_out/r05_gpu_suite_e.log 2>&1; tail -3 gpurun_out/r05_gpu_suite_e.log
for cfg in "--n 257 --rods 65280" "--n 129 --rods 130048" "--n 65 --rods 258048" "--n 273 --rods 61440" "--n 321 --rods 52224" "--n 130 --rods 130048 --operator 2"; do
  python tools/ab_apply.py $cfg --rounds 5 --reps 200 --libs lastcol=$PKG/libsri_hip.so chunk=$PKG/libsri_hip.so,SRI_DEBUG=lastcol=0 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r05_ab_lastcol.log 2>&1
cat gpurun_out/r05_ab_lastcol.log
