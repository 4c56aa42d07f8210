#!/bin/bash
# evidence_run.sh <tag> — the round's slower evidence in one gpurun call: long differential fuzz, soak, rod solves, many-panel and
# host-panel benches.  Writes gpurun_out/<tag>_*.{log,jsonl}.
tag=${1:-r04}
mkdir -p gpurun_out
( for seed in 11 12 13; do timeout -k 10 400 python tools/fuzz_apply.py --cases 500 --seed $seed 2>/dev/null | tail -4; done ) > gpurun_out/${tag}_fuzz_apply.log 2>&1; echo "fuzz rc=$?"
timeout -k 10 300 python tools/soak.py --reps 200 > gpurun_out/${tag}_soak.log 2>&1; echo "soak rc=$?"
: > gpurun_out/${tag}_rod_bench_final.jsonl
for spec in "16 262144" "17 262144" "32 262144" "33 131072" "48 65536" "64 65536" "128 131072" "256 65536" "300 49152" "512 16384" "1024 8192"; do
  set -- $spec
  timeout -k 10 200 python tools/rod_bench.py --n $1 --rods $2 --reps 5 2>/dev/null | tail -1 >> gpurun_out/${tag}_rod_bench_final.jsonl
done; echo "rod_bench done"
timeout -k 10 100 python tools/many_bench.py > gpurun_out/${tag}_many_bench.jsonl 2>/dev/null; timeout -k 10 100 python tools/many_bench.py --count 8 >> gpurun_out/${tag}_many_bench.jsonl 2>/dev/null; echo "many rc=$?"
timeout -k 10 200 python tools/host_panel_bench.py --pinned > gpurun_out/${tag}_host_panel_bench_final.jsonl 2>/dev/null
timeout -k 10 100 python tools/host_panel_bench.py --n 64 --rods 4096 >> gpurun_out/${tag}_host_panel_bench_final.jsonl 2>/dev/null
timeout -k 10 100 python tools/host_panel_bench.py --n 256 --rods 65536 --which 1 >> gpurun_out/${tag}_host_panel_bench_final.jsonl 2>/dev/null; echo "host rc=$?"
tail -3 gpurun_out/${tag}_fuzz_apply.log; tail -3 gpurun_out/${tag}_soak.log
