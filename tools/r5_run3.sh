#!/bin/bash
timeout -k 10 300 python tools/ab_step.py FUSE_DGRAD_REDUCE=0 TWIN_VIRTUAL=0 > gpurun_out/ab_step.log 2>&1
grep -v amdgpu.ids gpurun_out/ab_step.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=15 --deselect tests/test_gpu_dist.py > gpurun_out/gputest_b.log 2>&1
tail -40 gpurun_out/gputest_b.log
