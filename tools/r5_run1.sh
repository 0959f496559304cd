#!/bin/bash
# round 5: correctness of the two 16x16x32 kernels + same-box A/B against the 32x32x16 build (libonet_hip_old.so)
L=$PWD/onet_amd
ONET_HIP_LIB=$L/libonet_hip_p16.so timeout -k 10 400 python tools/pre16_check.py > gpurun_out/p16.log 2>&1
timeout -k 10 300 python tools/wgrad16_check.py > gpurun_out/w16.log 2>&1
: > gpurun_out/p16_ab.log; : > gpurun_out/w16_ab.log
for i in 1 2; do
ONET_HIP_LIB=$L/libonet_hip_old.so timeout -k 10 300 python tools/pre16_check.py time 2>&1 | grep "^lib\|^default" >> gpurun_out/p16_ab.log
ONET_HIP_LIB=$L/libonet_hip_p16.so timeout -k 10 300 python tools/pre16_check.py time 2>&1 | grep "^lib\|^default" >> gpurun_out/p16_ab.log
ONET_HIP_LIB=$L/libonet_hip_old.so timeout -k 10 300 python tools/wgrad16_check.py time 2>&1 | grep "^lib\|^default" >> gpurun_out/w16_ab.log
timeout -k 10 300 python tools/wgrad16_check.py time 2>&1 | grep "^lib\|^default" >> gpurun_out/w16_ab.log
done
grep -v amdgpu.ids gpurun_out/p16.log; grep -v amdgpu.ids gpurun_out/w16.log; cat gpurun_out/p16_ab.log gpurun_out/w16_ab.log
