#!/bin/bash
for v in "" _occ4; do
ONET_HIP_LIB=$PWD/onet_amd/libonet_hip$v.so timeout -k 10 300 python tools/time_z16.py 2>&1 | grep -v amdgpu.ids | cut -c1-250
done
