#!/bin/bash
# same-box A/B of an environment switch: tools/ab.sh VAR A_VALUE B_VALUE [pairs] [steps]  (run through gpurun)
VAR=$1; A=$2; B=$3; N=${4:-3}; S=${5:-20}
for i in $(seq $N); do
  for v in "$A" "$B"; do
    env $VAR=$v python bench.py --steps $S --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', d['ms_per_step'], d['roofline']['achieved'])" || exit 1
  done
done
