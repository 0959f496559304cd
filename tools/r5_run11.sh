#!/bin/bash
for v in "" _cp32; do
ONET_HIP_LIB=$PWD/onet_amd/libonet_hip$v.so timeout -k 10 400 python tools/convt_slots_check.py 64 2 2>&1 | grep -v amdgpu.ids | grep "^bwd\|wgrad sum" | sed "s/.*|| wgrad/[$v] wgrad/; s/| fp32.*//" 
done
