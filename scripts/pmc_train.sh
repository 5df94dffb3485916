#!/bin/bash
# HBM-traffic passes on the training step (run ON the MI355X box through gpurun): FETCH_SIZE and WRITE_SIZE of every kernel
# of 4 steps (scripts/profile_train.py), separate passes, kernel-trace only.  usage: bash scripts/pmc_train.sh <tag>; then
# python scripts/summarize_train_traffic.py <tag>  ->  profiles/<tag>_train_traffic.json
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_train_fetch -- python3 scripts/profile_train.py > $OUT/pmc_train_fetch.log 2>&1 || echo "fetch pass failed" >> $OUT/pmc_train_fail.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_train_write -- python3 scripts/profile_train.py > $OUT/pmc_train_write.log 2>&1 || echo "write pass failed" >> $OUT/pmc_train_fail.txt
echo done
