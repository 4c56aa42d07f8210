#!/bin/bash
# Round 5, re-entry: the rod table again with tools/rod_bench.py's corrected flop accounting (7 fine-grid columns where the
# closing product alone settles the solve; the earlier file counted round 4's 11).  Same library sources.
: > gpurun_out/r05_rod_bench_final.jsonl
for spec in "16 262144" "17 262144" "32 262144" "33 131072" "48 65536" "64 65536" "100 131072" "128 131072" "129 131072" "200 65536" "256 65536" "257 65536" "300 49152" "400 32768" "512 16384" "513 16384" "1024 8192"; do
  set -- $spec
  timeout -k 10 200 python tools/rod_bench.py --n $1 --rods $2 --reps 5 2>/dev/null | tail -1 >> gpurun_out/r05_rod_bench_final.jsonl
done; echo "rod_bench done"; wc -l gpurun_out/r05_rod_bench_final.jsonl
