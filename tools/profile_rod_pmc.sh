#!/bin/bash
# profile_rod_pmc.sh — PMC counters PER KERNEL of one batched rod solve (run on the GPU box via gpurun):
#   tools/profile_rod_pmc.sh <tag> [rod_bench.py arguments, default --n 256 --rods 65536]
#   pass 1: --kernel-trace --stats                      -> mean duration of each kernel of the solve
#   pass 2: --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
#   pass 3: --pmc FETCH_SIZE     pass 4: --pmc WRITE_SIZE   (own passes: TCC counter slots; never with runtime traces)
# Summary: profiles/<tag>_rod_pmc_by_kernel.json
set -e
tag=${1:-r05}
shift || true
ARGS="tools/rod_bench.py ${*:---n 256 --rods 65536} --reps 3 --no-cpu"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_rod_pmc_$tag
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $ARGS > $out/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/sq -- python $ARGS > $out/sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python $ARGS > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python $ARGS > $out/write.log 2>&1
python tools/summarize_rod_pmc.py $out $tag "$ARGS"
