#!/bin/bash
# Round 5, re-entry: evidence on the final sources (no source change): the coarse tolerance of the rod predictor far beyond the
# factors R5.8 tried (round 6's question: how loose may the coarse solve be), a fuzz run with a fresh seed, the rod table on one more box.
: > gpurun_out/r05_rod_ab_coarse_tol.log
for spec in "256 65536" "128 131072" "512 16384"; do
  set -- $spec
  for f in 2 32 512 8192 131072; do
    echo "N=$1 rods=$2 coarse_tol=$f x tol" >> gpurun_out/r05_rod_ab_coarse_tol.log
    SRI_DEBUG=coarse_tol=$f timeout -k 10 120 python tools/rod_bench.py --n $1 --rods $2 --reps 8 --no-cpu 2>/dev/null | tail -1 >> gpurun_out/r05_rod_ab_coarse_tol.log || exit 1
  done
done; echo "coarse_tol done"
timeout -k 10 400 python tools/fuzz_apply.py --cases 600 --seed 11 > gpurun_out/r05_fuzz_apply_seed11.log 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/r05_fuzz_apply_seed11.log
: > gpurun_out/r05_rod_bench_box2.jsonl
for spec in "64 65536" "128 131072" "256 65536" "257 65536" "512 16384" "1024 8192"; do
  set -- $spec
  timeout -k 10 200 python tools/rod_bench.py --n $1 --rods $2 --reps 5 2>/dev/null | tail -1 >> gpurun_out/r05_rod_bench_box2.jsonl
done; echo "rod_bench done"
