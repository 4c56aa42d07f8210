#!/bin/bash
# round-5 batch 15: EPI_QBC after the register diet (visible stores / loads, counted waits by the compiler)
python -m pytest tests/test_rod_gpu.py -m gpu -x -q -k "criterion or closing_product" > gpurun_out/r05_qbc_tests2.log 2>&1
rc=$?; tail -3 gpurun_out/r05_qbc_tests2.log
[ $rc -eq 0 ] || exit $rc
for n in 256:65536 512:16384; do
  python tools/rod_bench.py --n ${n%%:*} --rods ${n##*:} --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['n'], 'fused', d['fused_criterion'], d['ms_total'], d['last_update'])"
  python tools/rod_bench.py --n ${n%%:*} --rods ${n##*:} --no-cpu --no-fused-criterion 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['n'], 'fused', d['fused_criterion'], d['ms_total'], d['last_update'])"
done
