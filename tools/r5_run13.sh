#!/bin/bash
# configs[2] profile set refresh + bench
BENCH_ARGS="--conv bf16 --batch 256" bash tools/profile_round.sh r05c3 > gpurun_out/prof_r05c3.log 2>&1; tail -2 gpurun_out/prof_r05c3.log
bash tools/r5_run12.sh
