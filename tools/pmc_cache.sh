#!/bin/bash
# Texture-path / L2 counter passes over tools/pmc_kernels.py (run through gpurun from the repo root): WHICH=bf16 tools/pmc_cache.sh TAG
# latency of a TCP->TCC read = TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ; L2 hit rate = TCC_HIT / (TCC_HIT + TCC_MISS)
set -e
TAG=${1:-cache}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/tools/pmc_kernels.py"
P1="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum GRBM_GUI_ACTIVE SQ_WAVE_CYCLES"
P2="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum SQ_WAVE_CYCLES"
P3="TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TA_TA_BUSY_sum SQ_WAVE_CYCLES"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace -d $OUT/p$i -o c --output-format csv -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 $R/tools/pmc_diag_parse.py $OUT > $OUT/summary.txt 2>&1 || true
grep -n "== conv3x3\|== convt" -A 22 $OUT/summary.txt | grep -v "SQ_WAVE" | head -120
