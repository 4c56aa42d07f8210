#!/bin/bash
# profile_rod.sh — rocprofv3 evidence for the widened rows (run on the GPU box via gpurun):
#   pass 1: --kernel-trace --stats on the rod pipeline  -> per-kernel time of one batched solve
#   pass 2: --pmc SQ_LDS_BANK_CONFLICT ... on the step timing tool -> LDS conflicts of the fused step (permuted X read)
set -e
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_rod_$tag
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python tools/rod_bench.py --n 256 --rods 65536 --reps 3 --no-cpu > $out/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/lds -- python tools/step_bench.py --n 256 --rods 16384 --reps 20 --ramp 0.2 > $out/lds.log 2>&1
