#!/bin/bash
timeout -k 10 300 python tools/convt_slots_check.py 64 2 2>&1 | grep -v amdgpu.ids | sed "s/^/default /" | cut -c1-100
for v in occ3 occ4; do
ONET_HIP_LIB=$PWD/onet_amd/libonet_hip_$v.so timeout -k 10 300 python tools/convt_slots_check.py 64 2 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /" | cut -c1-100
done
ONET_HIP_LIB=$PWD/onet_amd/libonet_hip_occ4.so timeout -k 10 300 python tools/convt_slots_check.py 256 1 2>&1 | grep -v amdgpu.ids | sed "s/^/occ4 /" | cut -c1-100
