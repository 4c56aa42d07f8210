#!/bin/bash
# power_watch.sh — run a sustained bench in the background and sample power / clocks (measurement tooling).
out=${1:-gpurun_out/power.log}
python bench.py --steps 30000 --warmup 20 --cpu-seconds 0 --no-verify > gpurun_out/power_bench.json 2> gpurun_out/power_bench.err &
pid=$!
sleep 6
for i in $(seq 1 12); do
  echo "--- sample $i" >> $out
  rocm-smi --showpower --showclocks --showtemp --showperflevel 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temp|perf" >> $out
  sleep 0.4
done
wait $pid
rocm-smi --showmaxpower --showpower 2>/dev/null | grep -E "Power|Max" >> $out
cat gpurun_out/power_bench.json >> $out
