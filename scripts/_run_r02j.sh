set -e
mkdir -p gpurun_out/r02j
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash scripts/pmc_sq.sh r02j > gpurun_out/r02j/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r02j/pmc_fetch -- python3 scripts/pmc_scan.py > gpurun_out/r02j/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r02j/pmc_write -- python3 scripts/pmc_scan.py > gpurun_out/r02j/pmc_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02j/trace -- python3 bench.py --steps 5 --warmup 2 --no-graph --no-overlap --no-cpu-baseline --no-train --no-latency > gpurun_out/r02j/trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02j/trace_train -- python3 scripts/profile_train.py > gpurun_out/r02j/trace_train.log 2>&1
python scripts/measure_lowp_parity.py > gpurun_out/r02j/lowp.json 2> gpurun_out/r02j/lowp.err
