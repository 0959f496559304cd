#!/bin/bash
# full GPU suite with durations
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/gputest_f.log 2>&1
rc=$?
tail -40 gpurun_out/gputest_f.log
ROUNDS=3 timeout -k 10 300 python tools/ab_step.py CONVT_BWD_SLOTS=0 2>&1 | grep "^default\|^CONVT"
exit $rc
