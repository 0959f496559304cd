#!/bin/bash
# Collect the judged profile set on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh TAG      -> gpurun_out/prof_TAG/{stats,fetch,write,sq}/...
# kernel stats, then separate --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ busy counters), all of the same command.
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-secondary --no-f32-mfma-only --no-reference-loop $BENCH_ARGS"     # BENCH_ARGS e.g. "--conv bf16"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- $CMD > $OUT/stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- $CMD > $OUT/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- $CMD > $OUT/write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace -d $OUT/sq -o q --output-format csv -- $CMD > $OUT/sq.log 2>&1
grep -h '"metric"' $OUT/stats.log | tail -1 > $OUT/bench_line_under_profiler.json || true
ls $OUT
