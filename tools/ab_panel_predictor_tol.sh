#!/bin/bash
# As ab_panel_predictor.sh, over the tolerance of the fixed point: where does a start that does not contain the fine grid's own
# operator deviation (~2e-14, smooth) still settle in one sweep?  Output: gpurun_out/r05_rod_ab_panel_predictor_tol.jsonl
# NEEDS tools/studies/panel_predictor_r05.patch applied and the library rebuilt (EXPERIMENTS R5.12: the option is not in the product).
set -e
cd "$(dirname "$0")/.."
out=gpurun_out/r05_rod_ab_panel_predictor_tol.jsonl
: > $out
for cfg in "256 65536" "128 131072"; do
  set -- $cfg
  for tol in 3e-14 1e-13 1e-12 1e-10; do
    python3 tools/rod_bench.py --no-cpu --reps 20 --n $1 --rods $2 --tol $tol >> $out
    python3 tools/rod_bench.py --no-cpu --reps 20 --n $1 --rods $2 --tol $tol --no-panel-predictor >> $out
  done
done
