#!/bin/bash
tag=r05
: > gpurun_out/${tag}_rod_bench_final.jsonl
for spec in "16 262144" "17 262144" "32 262144" "33 131072" "48 65536" "64 65536" "100 131072" "128 131072" "129 131072" "200 65536" "256 65536" "257 65536" "300 49152" "400 32768" "512 16384" "513 16384" "1024 8192"; do
  set -- $spec
  timeout -k 10 200 python tools/rod_bench.py --n $1 --rods $2 --reps 5 2>/dev/null | tail -1 >> gpurun_out/${tag}_rod_bench_final.jsonl
done; echo "rod_bench done"
timeout -k 10 100 python tools/many_bench.py > gpurun_out/${tag}_many_bench.jsonl 2>/dev/null; timeout -k 10 100 python tools/many_bench.py --count 8 >> gpurun_out/${tag}_many_bench.jsonl 2>/dev/null; echo "many rc=$?"
timeout -k 10 200 python tools/host_panel_bench.py --pinned > gpurun_out/${tag}_host_panel_bench_final.jsonl 2>/dev/null
timeout -k 10 100 python tools/host_panel_bench.py --n 256 --rods 65536 --which 1 >> gpurun_out/${tag}_host_panel_bench_final.jsonl 2>/dev/null; echo "host rc=$?"
python tools/column_plan_probe.py --rounds 3 --reps 60 > gpurun_out/${tag}_column_plan_probe_final.log 2>&1; echo "plan probe rc=$?"
