#!/bin/bash
# SQ counter passes over tools/pmc_kernels.py (run through gpurun from the repo root): tools/pmc_diag.sh TAG
set -e
TAG=${1:-diag}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/tools/${PMC_SCRIPT:-pmc_kernels.py}"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE"
P3="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL"
P4="SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace -d $OUT/p$i -o c --output-format csv -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 $R/tools/pmc_diag_parse.py $OUT > $OUT/summary.txt 2>&1 || true
tail -50 $OUT/summary.txt
