#!/bin/bash
# rod solve with / without the first-sweep check
for cfg in "--n 256 --rods 65536" "--n 128 --rods 131072" "--n 300 --rods 49152" "--n 512 --rods 16384" "--n 1024 --rods 8192" "--n 256 --rods 65536 --tol 1e-13"; do
  python tools/rod_bench.py $cfg --no-cpu || exit 1
  python tools/rod_bench.py $cfg --no-cpu --no-first-sweep-check || exit 1
done
