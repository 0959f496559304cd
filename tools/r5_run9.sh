#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "repack or convT or weight_packs" > gpurun_out/t9a.log 2>&1; tail -5 gpurun_out/t9a.log
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x > gpurun_out/t9b.log 2>&1; tail -8 gpurun_out/t9b.log
ROUNDS=3 timeout -k 10 300 python tools/ab_step.py PACK_MULTI=0 2>&1 | grep "^default\|^PACK"
B=256 CONV=bf16 ROUNDS=2 STEPS=4 timeout -k 10 300 python tools/ab_step.py PACK_MULTI=0 2>&1 | grep "^default\|^PACK" | sed 's/^/c3 /'
