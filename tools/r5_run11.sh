#!/bin/bash
timeout -k 10 300 python tools/time_z16.py 2>&1 | grep -v amdgpu.ids | cut -c1-250
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "presplit_producers or bn_relu or pool" 2>&1 | tail -3
timeout -k 10 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_gradients.py -m gpu -q -x -k "bf16 or config3 or golden" 2>&1 | tail -3
B=256 CONV=bf16 ROUNDS=2 STEPS=4 timeout -k 10 300 python tools/ab_step.py 2>&1 | grep "^default" | sed 's/^/c3 /'
ROUNDS=3 timeout -k 10 300 python tools/ab_step.py 2>&1 | grep "^default"
