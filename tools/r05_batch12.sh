#!/bin/bash
PKG=experimental_gpu_programming_for_a_spectral_numerical_integration_amd
python -m pytest tests/test_apply_gpu.py -m gpu -x -q > gpurun_out/r05_gpu_suite_i.log 2>&1; tail -3 gpurun_out/r05_gpu_suite_i.log
for cfg in "--n 257 --rods 65280" "--n 513 --rods 32704" "--n 1025 --rods 16352" "--n 258 --rods 65024 --operator 1" "--n 260 --rods 64512" "--n 1028 --rods 16320 --operator 2"; do
  python tools/ab_apply.py $cfg --rounds 5 --reps 200 --libs mini=$PKG/libsri_hip.so strip=$PKG/libsri_hip.so,SRI_DEBUG=mini=0 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r05_ab_mini_strip.log 2>&1
cat gpurun_out/r05_ab_mini_strip.log
