#!/bin/bash
tag=r05
( for seed in 21 22; do timeout -k 10 420 python tools/fuzz_apply.py --cases 500 --seed $seed 2>/dev/null | tail -4; done ) > gpurun_out/${tag}_fuzz_apply.log 2>&1; echo "fuzz rc=$?"
timeout -k 10 300 python tools/soak.py --reps 200 > gpurun_out/${tag}_soak.log 2>&1; echo "soak rc=$?"
tail -3 gpurun_out/${tag}_fuzz_apply.log; tail -3 gpurun_out/${tag}_soak.log
