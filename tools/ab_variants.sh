#!/bin/bash
# same-box A/B of library variants (onet_amd.build --variant): tools/ab_variants.sh "base asym noslp" [rounds] [steps]
# prints ms/step and the dominant kernel's numbers per run  (run through gpurun from the repo root)
VARS=${1:-base}; N=${2:-2}; S=${3:-20}
for i in $(seq $N); do
  for v in $VARS; do
    if [ "$v" = base ]; then LIBV=""; else LIBV="$PWD/onet_amd/libonet_hip_$v.so"; fi
    ONET_HIP_LIB=$LIBV python bench.py --steps $S --warmup 3 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$v', d['ms_per_step'], ' '.join('%s=%.3fms/%.0fTF' % (n.replace('conv_','').replace('_kernel',''), v['avg_ms'], v['direct_equivalent_tflops']) for n, v in k.items()), flush=True)" || exit 1
  done
done
