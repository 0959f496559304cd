#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "second_forward or golden or twin" > gpurun_out/t9b.log 2>&1; tail -6 gpurun_out/t9b.log
