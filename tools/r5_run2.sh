#!/bin/bash
# round 5: pre16 kernel (epilogue with whole-line stores) vs the 32x32x16 build, same box, interleaved
L=$PWD/onet_amd
ONET_HIP_LIB=$L/libonet_hip_p16.so timeout -k 10 400 python tools/pre16_check.py > gpurun_out/p16.log 2>&1
: > gpurun_out/p16_ab.log
for i in 1 2 3; do
ONET_HIP_LIB=$L/libonet_hip_old.so timeout -k 10 300 python tools/pre16_check.py time 2>&1 | grep "^lib\|^default" >> gpurun_out/p16_ab.log
ONET_HIP_LIB=$L/libonet_hip_p16.so timeout -k 10 300 python tools/pre16_check.py time 2>&1 | grep "^lib\|^default" >> gpurun_out/p16_ab.log
done
grep -v amdgpu.ids gpurun_out/p16.log | tail -3; cat gpurun_out/p16_ab.log
