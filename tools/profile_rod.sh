#!/bin/bash
# profile_rod.sh — rocprofv3 evidence for the widened rows (run on the GPU box via gpurun):
#   pass 1: --kernel-trace --stats on the rod pipeline  -> per-kernel time of one batched solve
#   pass 2: --pmc SQ_LDS_BANK_CONFLICT ... on the step timing tool -> LDS conflicts / MFMA busy of the fused step
#   pass 3: --pmc FETCH_SIZE, pass 4: --pmc WRITE_SIZE on the step timing tool -> HBM traffic of the fused Picard step
#           (separate passes: TCC has 4 counter slots; never combined with runtime traces)
# Summary: profiles/<tag>_rod_rocprofv3_kernel_stats.csv, profiles/<tag>_fused_step_pmc.json
set -e
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_rod_$tag
rm -rf $out && mkdir -p $out
STEP="tools/step_bench.py --n 256 --rods 65536 --reps 20 --ramp 0.2"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python tools/rod_bench.py --n 256 --rods 65536 --reps 3 --no-cpu > $out/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/lds -- python $STEP > $out/lds.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python $STEP > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python $STEP > $out/write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/steptrace -- python $STEP > $out/steptrace.log 2>&1
python tools/summarize_rod_profile.py $out $tag
