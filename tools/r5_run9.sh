#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "convT or presplit_upper" > gpurun_out/t9a.log 2>&1; tail -6 gpurun_out/t9a.log
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_gradients.py -m gpu -q -x -k "bf16 or config3" > gpurun_out/t9b.log 2>&1; tail -8 gpurun_out/t9b.log
B=256 CONV=bf16 ROUNDS=2 STEPS=4 timeout -k 10 300 python tools/ab_step.py CONVT_BWD_SLOTS=0 2>&1 | grep "^default\|^CONVT" | sed 's/^/c3 /'
