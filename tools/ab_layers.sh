#!/bin/bash
# same-box A/B of per-layer kernel timings: default library vs variant builds (onet_amd/libonet_hip_<v>.so), interleaved rounds
# usage (through gpurun): bash tools/ab_layers.sh tools/time_split_pre_layers.py 3 m16 [more variants]
script=$1; rounds=$2; shift 2
for i in $(seq $rounds); do
  python $script || exit 1
  for v in "$@"; do ONET_HIP_LIB=$PWD/onet_amd/libonet_hip_$v.so python $script || exit 1; done
done
