set -e
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu.txt 2>&1 || (tail -40 gpurun_out/r02/pytest_gpu.txt; exit 1)
tail -3 gpurun_out/r02/pytest_gpu.txt
bash scripts/collect_profiles.sh r02 > gpurun_out/r02/collect.log 2>&1
bash scripts/pmc_sq.sh r02 > gpurun_out/r02/pmc_sq.log 2>&1
python scripts/measure_lowp_parity.py > gpurun_out/r02/lowp.json 2> gpurun_out/r02/lowp.err
