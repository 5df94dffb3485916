#!/bin/bash
# BASELINE configs[4] (Tramba-V 768x768 fp16, batch 2: L = 36864 on the 192x192 maps) -- bench line + kernel trace, run ON the
# MI355X box through gpurun: bash scripts/collect_config5.sh <tag>; then python scripts/summarize_config5.py <tag>
set -o pipefail
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --img 768 --dtype fp16 --batch 2 --steps 30 --warmup 10 --no-train --no-cpu-baseline > $OUT/c5_bench.json 2> $OUT/c5_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5_trace -- python3 bench.py --img 768 --dtype fp16 --batch 2 --steps 5 --warmup 2 --no-graph --no-overlap --no-cpu-baseline --no-train --no-latency > $OUT/c5_trace.log 2>&1
python scripts/trace_by_grid.py $OUT/c5_trace ss2d > $OUT/c5_scan_by_grid.txt
python scripts/trace_by_grid.py $OUT/c5_trace selective_scan >> $OUT/c5_scan_by_grid.txt
echo done
