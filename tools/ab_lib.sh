#!/bin/bash
# same-box A/B of two builds of the library: onet_amd/libonet_prev.bin vs onet_amd/libonet_new.bin  (run through gpurun)
N=${1:-3}; S=${2:-20}
for i in $(seq $N); do
  for v in prev new; do
    cp onet_amd/libonet_$v.bin onet_amd/libonet_hip.so
    python bench.py --steps $S --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['roofline']['achieved'])" || exit 1
  done
done
