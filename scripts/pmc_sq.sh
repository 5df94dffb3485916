#!/bin/bash
# SQ-counter passes on the Helix-SS2D pair (run ON the MI355X box through gpurun): is the scan issue-bound?
# usage: bash scripts/pmc_sq.sh <tag>
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L > $OUT/counters_list.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAIT_INST_LDS" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/sq$i -- python3 scripts/pmc_scan.py helix > $OUT/sq$i.log 2>&1 || echo "pass $i failed" >> $OUT/sq_fail.txt
done
echo done
