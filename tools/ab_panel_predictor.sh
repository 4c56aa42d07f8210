#!/bin/bash
# A/B of the rod solve's predicted start: four 16-node panels (default) against one 32-node interval (--no-panel-predictor),
# same process order on one box, tools/rod_bench.py.  Output: gpurun_out/r05_rod_ab_panel_predictor.jsonl
# NEEDS tools/studies/panel_predictor_r05.patch applied and the library rebuilt (EXPERIMENTS R5.12: the option is not in the product).
set -e
cd "$(dirname "$0")/.."
out=gpurun_out/r05_rod_ab_panel_predictor.jsonl
: > $out
for cfg in "256 65536" "128 131072" "512 32768" "1024 8192" "96 65536" "300 32768"; do
  set -- $cfg
  for rep in 1 2; do
    python3 tools/rod_bench.py --no-cpu --reps 20 --n $1 --rods $2 >> $out
    python3 tools/rod_bench.py --no-cpu --reps 20 --n $1 --rods $2 --no-panel-predictor >> $out
  done
done
