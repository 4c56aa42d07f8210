#!/bin/bash
# Prices the panel predictor of EXPERIMENTS R5.11 with what the library has today: a panel of a quarter (half) of the rod with
# strain K is the same fixed point as a whole rod with strain K/4 (K/2), so the on-chip N = 16 solve of 4 x (2 x) the rods at
# `--scale 0.5` (`1.0`) IS the work of four (two) 16-node panels per rod — sweep count and time — plus positions and both output
# panels, which a predictor would not write (an upper bound).  Reference rows on the same box: today's solves at N = 256 / 128.
set -e
cd "$(dirname "$0")/../.."
out=gpurun_out/r05_panel_predictor_price.jsonl
: > $out
run() { python3 tools/rod_bench.py --no-cpu --reps 20 "$@" >> $out; }
run --n 256 --rods 65536
run --n 128 --rods 131072
run --n 16 --rods 262144 --scale 0.5
run --n 16 --rods 262144 --scale 2.0
run --n 16 --rods 196608 --scale 0.6667
run --n 16 --rods 131072 --scale 1.0
run --n 16 --rods 524288 --scale 0.5
run --n 16 --rods 65536 --scale 2.0
run --n 32 --rods 65536 --scale 2.0
run --n 32 --rods 131072 --scale 1.0
for t in 1e-3 1e-6 1e-9 1e-12; do run --n 16 --rods 262144 --scale 0.5 --tol $t; done
cat $out | cut -c1-600
