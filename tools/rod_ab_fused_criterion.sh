#!/bin/bash
# rod_ab_fused_criterion.sh — the rod GPU tests, then the solve with / without the criterion formed in the closing product (EPI_QBC)
python -m pytest tests/test_rod_gpu.py -m gpu -x -q > gpurun_out/r05_rod_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r05_rod_tests.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
for n in 100:131072 128:131072 129:131072 256:65536 257:65536 300:49152 400:32768 512:16384 513:16384 1024:8192; do
  for f in "" "--no-fused-criterion"; do
    python tools/rod_bench.py --n ${n%%:*} --rods ${n##*:} --no-cpu $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['n'], d['rods'], 'fused_criterion', d['fused_criterion'], round(d['ms_total'],4), d['last_update'])" >> gpurun_out/r05_rod_ab_fused_criterion.log
  done
done
done
cat gpurun_out/r05_rod_ab_fused_criterion.log
