#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_gradients.py -m gpu -q -x -k "golden or benchmark_dispatch or presplit or config3 or frozen or 8x8 or b8_c1_128-auto or fresh_seed or train" > gpurun_out/t9b.log 2>&1; tail -8 gpurun_out/t9b.log
ROUNDS=3 timeout -k 10 300 python tools/ab_step.py CONVT_BWD_SLOTS=0 CONVT_SLOTS=0 2>&1 | grep "^default\|^CONVT"
