#!/bin/bash
timeout -k 10 1100 python -m pytest tests/test_gpu_gradients.py tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -q --durations=15 -k "not routed_fp64_oracle and not benchmark_dispatch and not two_pass_and_unshared" > gpurun_out/gputest_c.log 2>&1
tail -60 gpurun_out/gputest_c.log
