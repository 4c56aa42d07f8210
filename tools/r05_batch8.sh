#!/bin/bash
python -m pytest tests -m gpu -x -q > gpurun_out/r05_gpu_suite_f.log 2>&1; tail -3 gpurun_out/r05_gpu_suite_f.log
for i in 1 2 3; do
  python tools/rod_bench.py --n 256 --rods 65536 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nt   ', d['ms_total'], d['last_update'])"
  SRI_HIP_LIBRARY=$PWD/build/libsri_hip_confirmplain.so python tools/rod_bench.py --n 256 --rods 65536 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain', d['ms_total'], d['last_update'])"
done
