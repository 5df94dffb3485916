#!/bin/bash
# SQ counter passes (a TA / TCP pass aborted rocprofv3 on this image and is not made) on the streaming kernels of scripts/dev/run_stream_kernels.py (run ON the MI355X box through gpurun)
# usage: bash scripts/dev/pmc_stream.sh <tag> [scanbwd];  then python scripts/dev/summarize_stream.py <tag>
TAG=${1:-r04}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/st$i -- python3 scripts/dev/run_stream_kernels.py $2 > $OUT/st$i.log 2>&1 || echo "pass $i failed" >> $OUT/st_fail.txt
done
echo done
