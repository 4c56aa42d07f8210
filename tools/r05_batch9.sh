#!/bin/bash
python -m pytest tests/test_rod_gpu.py tests/test_examples_gpu.py -m gpu -x -q > gpurun_out/r05_gpu_suite_g.log 2>&1; tail -3 gpurun_out/r05_gpu_suite_g.log
cat > /tmp/rodab.py <<'PY'
import sys, json, subprocess
PY
for cfg in "--n 256 --rods 65536" "--n 128 --rods 131072" "--n 512 --rods 16384" "--n 1024 --rods 8192" "--n 300 --rods 49152"; do
  for i in 1 2; do
    python tools/rod_bench.py $cfg --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('q+b  ', d['n'], d['ms_total'], d['last_update'])"
    python tools/rod_bench.py $cfg --no-cpu --no-fused-position-rhs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pass ', d['n'], d['ms_total'], d['last_update'])"
  done
done
