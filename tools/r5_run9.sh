#!/bin/bash
for i in 1 2; do
ONET_HIP_LIB=$PWD/onet_amd/libonet_hip_oldpack.so ROUNDS=3 timeout -k 10 300 python tools/ab_step.py 2>&1 | grep "^default" | sed 's/^default/oldpack/'
ROUNDS=3 timeout -k 10 300 python tools/ab_step.py 2>&1 | grep "^default" | sed 's/^default/newpack/'
done
