#!/bin/bash
# same-box A/B of two library builds through the whole benchmark step (two processes, interleaved)
L=$PWD/onet_amd
for i in 1 2 3; do
  ONET_HIP_LIB=$L/libonet_hip_rdold.so ROUNDS=3 timeout -k 10 200 python tools/ab_step.py 2>&1 | grep "^default" | sed 's/^default/rdold  /'
  ROUNDS=3 timeout -k 10 200 python tools/ab_step.py 2>&1 | grep "^default" | sed 's/^default/rdnew  /'
done
ONET_HIP_LIB=$L/libonet_hip_rdold.so B=256 CONV=bf16 ROUNDS=2 STEPS=4 timeout -k 10 300 python tools/ab_step.py 2>&1 | grep "^default" | sed 's/^default/c3 rdold/'
B=256 CONV=bf16 ROUNDS=2 STEPS=4 timeout -k 10 300 python tools/ab_step.py FUSE_DGRAD_REDUCE=0 2>&1 | grep "^default\|^FUSE" | sed 's/^/c3 rdnew /'
