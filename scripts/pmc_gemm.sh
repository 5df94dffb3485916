#!/bin/bash
# SQ-counter passes on the five heaviest GEMM shapes (run ON the MI355X box through gpurun): MFMA utilisation of the projections.
# usage: bash scripts/pmc_gemm.sh <tag>   then   python scripts/summarize_gemm_counters.py <tag>
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/gemm_sq$i -- python3 scripts/pmc_gemm5.py > $OUT/gemm_sq$i.log 2>&1 || echo "pass $i failed" >> $OUT/gemm_sq_fail.txt
done
echo done
