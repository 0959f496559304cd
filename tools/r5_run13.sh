#!/bin/bash
# the judged profile sets of the round (headline and configs[2]) + the per-launch table
bash tools/profile_round.sh r05 > gpurun_out/prof_r05.log 2>&1; tail -2 gpurun_out/prof_r05.log
BENCH_ARGS="--conv bf16 --batch 256" bash tools/profile_round.sh r05c3 > gpurun_out/prof_r05c3.log 2>&1; tail -2 gpurun_out/prof_r05c3.log
timeout -k 10 300 python tools/step_layer_table.py 10 > gpurun_out/steptab_r05.md 2> gpurun_out/steptab_r05.err; wc -l gpurun_out/steptab_r05.md
