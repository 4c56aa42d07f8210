#!/bin/bash
PKG=experimental_gpu_programming_for_a_spectral_numerical_integration_amd
python -m pytest tests -m gpu -x -q > gpurun_out/r05_gpu_suite_d.log 2>&1; tail -3 gpurun_out/r05_gpu_suite_d.log
for i in 1 2; do
  python tools/step_bench.py --n 256 --reps 600 > gpurun_out/r05_step_late_$i.json 2>/dev/null
  SRI_HIP_LIBRARY=$PWD/build/libsri_hip_earlyread.so python tools/step_bench.py --n 256 --reps 600 > gpurun_out/r05_step_early_$i.json 2>/dev/null
done
python - <<'PY'
import json
for tag in ("late_1","early_1","late_2","early_2"):
    d=json.load(open(f"gpurun_out/r05_step_{tag}.json"))
    print(tag, " ".join(f"{k}={v['ms']:.4f}" for k,v in d.items() if isinstance(v,dict)))
PY
python tools/rod_bench.py --n 256 --rods 65536 --no-cpu | cut -c1-400
python tools/rod_bench.py --n 1024 --rods 8192 --no-cpu | cut -c1-400
