#!/bin/bash
# round-5 batch: full GPU suite, fused-step bench, column plan probe (own / model / measured)
python -m pytest tests -m gpu -x -q > gpurun_out/r05_gpu_suite_c.log 2>&1; tail -3 gpurun_out/r05_gpu_suite_c.log
python tools/step_bench.py --n 256 > gpurun_out/r05_step_bench_b.json 2>/dev/null
python tools/column_plan_probe.py --rounds 3 --reps 60 > gpurun_out/r05_column_plan_probe.log 2>&1; tail -5 gpurun_out/r05_column_plan_probe.log
