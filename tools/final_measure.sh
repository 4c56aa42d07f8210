#!/bin/bash
# Round-end measurements on the GPU box: bench lines, rocprofv3 summaries, step / rod / parity reports.
#   tools/final_measure.sh <tag> [a|b|all]   -> gpurun_out/<tag>_* and profiles/<tag>_* (copy what you want judged into profiles/)
# Two halves so that each fits one gpurun call: a = profiles + bench lines of configs 3 / 2 / 5;  b = the 255-row operator
# (ld = 255 and 256), step / parity / rod reports.  The profile passes come first: they refresh the PMC traffic records that the
# bench lines replay (keyed on the library's source hash: a line taken before its record was refreshed says traffic_stale).
#   tools/final_measure.sh <tag> lines   -> only the six bench lines (after profiles/ holds records of the current sources)
tag=${1:-r04}
part=${2:-all}
mkdir -p gpurun_out
touch gpurun_out/.final_measure_start
if [ "$part" = "a" ] || [ "$part" = "all" ]; then
PROFILE_STEPS=200 tools/profile_bench.sh ${tag}_cfg3 > gpurun_out/prof_${tag}_cfg3.log 2>&1; echo "profile cfg3 rc=$?"
PROFILE_STEPS=5000 tools/profile_bench.sh ${tag}_cfg2 --config 2 > gpurun_out/prof_${tag}_cfg2.log 2>&1; echo "profile cfg2 rc=$?"
PROFILE_STEPS=200 tools/profile_bench.sh ${tag}_cfg5 --config 5 > gpurun_out/prof_${tag}_cfg5.log 2>&1; echo "profile cfg5 rc=$?"
python bench.py > gpurun_out/${tag}_bench_n256.json 2> gpurun_out/${tag}_bench_n256.err; echo "bench cfg3 rc=$?"
python bench.py --config 2 --steps 20000 --warmup 1000 > gpurun_out/${tag}_bench_cfg2.json 2>/dev/null; echo "bench cfg2 rc=$?"
python bench.py --config 5 --steps 1000 > gpurun_out/${tag}_bench_cfg5.json 2>/dev/null; echo "bench cfg5 rc=$?"
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_driver_form.json 2>/dev/null; echo "bench driver form rc=$?"
fi
if [ "$part" = "c" ]; then
# round 5: the two grid sizes between the BASELINE configs that sit furthest below their roofline, with the prescribed PMC passes
PROFILE_STEPS=200 tools/profile_bench.sh ${tag}_n128 --n 128 --rods 131072 > gpurun_out/prof_${tag}_n128.log 2>&1; echo "profile n128 rc=$?"
PROFILE_STEPS=200 tools/profile_bench.sh ${tag}_n96 --n 96 --rods 174752 > gpurun_out/prof_${tag}_n96.log 2>&1; echo "profile n96 rc=$?"
bash tools/sweep_n.sh ${tag} > gpurun_out/sweep_${tag}.log 2>&1; echo "sweep rc=$?"
fi
if [ "$part" = "lines" ]; then
python bench.py > gpurun_out/${tag}_bench_n256.json 2> gpurun_out/${tag}_bench_n256.err; echo "bench cfg3 rc=$?"
python bench.py --config 2 --steps 20000 --warmup 1000 > gpurun_out/${tag}_bench_cfg2.json 2>/dev/null; echo "bench cfg2 rc=$?"
python bench.py --config 5 --steps 1000 > gpurun_out/${tag}_bench_cfg5.json 2>/dev/null; echo "bench cfg5 rc=$?"
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_driver_form.json 2>/dev/null; echo "bench driver form rc=$?"
python bench.py --operator dnn_inv > gpurun_out/${tag}_bench_dnn_inv_255.json 2>/dev/null; echo "bench dnn_inv rc=$?"
python bench.py --operator dnn_inv --ld 256 > gpurun_out/${tag}_bench_dnn_inv_255_ld256.json 2>/dev/null; echo "bench dnn_inv ld 256 rc=$?"
fi
if [ "$part" = "b" ] || [ "$part" = "all" ]; then
PROFILE_STEPS=200 tools/profile_bench.sh ${tag}_dnn_inv_255 --operator dnn_inv > gpurun_out/prof_${tag}_255.log 2>&1; echo "profile 255 rc=$?"
PROFILE_STEPS=200 tools/profile_bench.sh ${tag}_dnn_inv_255_ld256 --operator dnn_inv --ld 256 > gpurun_out/prof_${tag}_255_ld256.log 2>&1; echo "profile 255 ld 256 rc=$?"
python bench.py --operator dnn_inv > gpurun_out/${tag}_bench_dnn_inv_255.json 2>/dev/null; echo "bench dnn_inv rc=$?"
python bench.py --operator dnn_inv --ld 256 > gpurun_out/${tag}_bench_dnn_inv_255_ld256.json 2>/dev/null; echo "bench dnn_inv ld 256 rc=$?"
python tools/step_bench.py > gpurun_out/${tag}_step_bench.json 2>/dev/null; echo "step_bench rc=$?"
python tools/parity_report.py > gpurun_out/${tag}_parity_report.json 2>/dev/null; echo "parity_report rc=$?"
tools/profile_rod.sh ${tag} > gpurun_out/prof_rod_${tag}.log 2>&1; echo "profile_rod rc=$?"
fi
# only what THIS run wrote into profiles/ on the box (summaries, traffic records): a committed file of the same name that this run
# did not refresh must never travel back over a fresh one
find profiles -maxdepth 1 -type f -newer gpurun_out/.final_measure_start -exec cp {} gpurun_out/ \;
grep -h -o '"frac": [0-9.]*' gpurun_out/${tag}_bench_*.json
