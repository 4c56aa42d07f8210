#!/bin/bash
# round-5 batch 14: the criterion in the closing product (EPI_QBC) — tests, then the rod solve with and without it
python -m pytest tests/test_rod_gpu.py -m gpu -x -q -k "criterion or closing_product or first_sweep" > gpurun_out/r05_qbc_tests.log 2>&1
rc=$?; tail -15 gpurun_out/r05_qbc_tests.log
[ $rc -eq 0 ] || exit $rc
python tools/rod_bench.py --n 256 --rods 65536 --no-cpu > gpurun_out/r05_qbc_rod_256.json 2>gpurun_out/r05_qbc_rod_256.err && cat gpurun_out/r05_qbc_rod_256.json
python tools/rod_bench.py --n 512 --rods 16384 --no-cpu > gpurun_out/r05_qbc_rod_512.json 2>gpurun_out/r05_qbc_rod_512.err && cat gpurun_out/r05_qbc_rod_512.json
python tools/rod_bench.py --n 256 --rods 65536 --no-fused-criterion --no-cpu > gpurun_out/r05_qbc_rod_256_off.json 2>/dev/null && cat gpurun_out/r05_qbc_rod_256_off.json
python tools/rod_bench.py --n 512 --rods 16384 --no-fused-criterion --no-cpu > gpurun_out/r05_qbc_rod_512_off.json 2>/dev/null && cat gpurun_out/r05_qbc_rod_512_off.json
