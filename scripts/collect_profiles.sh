#!/bin/bash
# Run ON the MI355X box (through gpurun) from the repo root: collects the rocprofv3 summaries that
# profiles/ keeps.  Counter passes are separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 30 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err
# (bench.py --no-overlap: one stream, per-kernel durations of kernels running alone)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 5 --warmup 2 --no-graph --no-overlap --no-cpu-baseline --no-train --no-latency > $OUT/trace.log 2>&1

rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 scripts/pmc_scan.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 scripts/pmc_scan.py > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_train -- python3 scripts/profile_train.py > $OUT/trace_train.log 2>&1
echo done
