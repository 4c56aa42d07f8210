#!/bin/bash
# profile_bench.sh — rocprofv3 evidence for bench.py's roofline object (run on the GPU box via gpurun).
#   tools/profile_bench.sh <tag> [bench.py arguments, e.g. --config 5]
#   pass 1: --kernel-trace --stats            -> average duration of the apply kernel
#   pass 2: --pmc FETCH_SIZE                  -> HBM read bytes   (own pass: TCC has 4 counter slots, FETCH_SIZE takes 3)
#   pass 3: --pmc WRITE_SIZE                  -> HBM write bytes
#   pass 4: --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -> MFMA utilisation
# PMC passes never use --sys-trace / runtime traces (gpurun refuses that combination).
# Summaries land in profiles/<tag>_rocprofv3_{kernel_stats.csv,summary.json} and profiles/hbm_traffic_n<N>_r<rods>_<op>.json.
set -e
tag=${1:-r03}
shift || true
# The profiled program must be the single-rank bench.py itself: with --gpus N > 1 bench.py starts torch.distributed.run as
# a child of a process whose GPU the profiler's preloaded library has already initialised (in the --pmc passes) — the
# launcher hop that must never sit behind `rocprofv3 --`.  Multi-GPU evidence comes from unprofiled runs.
prev=""
for a in "$@"; do
  if [ "$prev" = "--gpus" ] && [ "$a" != "1" ]; then echo "profile_bench.sh: --gpus $a refused (profile the single-rank bench only)" >&2; exit 2; fi
  case "$a" in --gpus=1) ;; --gpus=*) echo "profile_bench.sh: $a refused (profile the single-rank bench only)" >&2; exit 2;; esac
  prev="$a"
done
if [ -n "$WORLD_SIZE" ] && [ "$WORLD_SIZE" != "1" ]; then echo "profile_bench.sh: WORLD_SIZE=$WORLD_SIZE refused" >&2; exit 2; fi
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
ARGS="bench.py --steps ${PROFILE_STEPS:-200} --warmup 20 --ramp-ms 800 --cpu-seconds 0 --no-verify $*"
python $ARGS > $out/bench_line.json 2> $out/bench_line.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $ARGS > $out/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python $ARGS > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python $ARGS > $out/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/sq -- python $ARGS > $out/sq.log 2>&1 || true
python tools/summarize_profile.py $out $tag
