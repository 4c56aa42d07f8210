#!/bin/bash
# rod_ab_criterion_geometry.sh — the closing product with the criterion (EPI_QBC) of operators whose own geometry is the 256-row one:
# on the handle's 128-row packing (default) or in the own geometry (SRI_DEBUG=criterion_geometry=own); rod tests first.
python -m pytest tests/test_rod_gpu.py -m gpu -x -q > gpurun_out/r05_rod_tests7.log 2>&1
rc=$?; tail -3 gpurun_out/r05_rod_tests7.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
for n in 256:65536 257:65536 400:32768 512:16384 513:16384 1024:8192 200:65536; do
  for e in "" "criterion_geometry=own"; do
    SRI_DEBUG=$e python tools/rod_bench.py --n ${n%%:*} --rods ${n##*:} --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['n'], d['rods'], '${e:-128-row tiles}', round(d['ms_total'],4), d['last_update'])" >> gpurun_out/r05_rod_ab_criterion_geometry.log
  done
done
done
cat gpurun_out/r05_rod_ab_criterion_geometry.log
