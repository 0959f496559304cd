#!/bin/bash
# full GPU suite with durations
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/gputest_f.log 2>&1
rc=$?
tail -25 gpurun_out/gputest_f.log
exit $rc
