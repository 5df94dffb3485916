set -e
mkdir -p gpurun_out/r02c
python scripts/bench_scan.py helix0 helix1 helix2 > gpurun_out/r02c/bench_scan.txt 2>&1
bash scripts/pmc_sq.sh r02c > gpurun_out/r02c/pmc.log 2>&1
python bench.py > gpurun_out/r02c/bench.json 2> gpurun_out/r02c/bench.err
python bench.py --gpus 2 --backend gloo --steps 4 --warmup 1 --no-latency --no-cpu-baseline > gpurun_out/r02c/bench_g2.json 2> gpurun_out/r02c/bench_g2.err || echo "gloo-2 bench failed" >> gpurun_out/r02c/fail.txt
python -m pytest tests/test_gpu_model.py tests/test_gpu_e2e.py -x -q -k "elementwise or low_precision or tramba_r_384 or loss_on_device or rccl or current_weights or graphed_train" > gpurun_out/r02c/pytest.txt 2>&1 || (tail -40 gpurun_out/r02c/pytest.txt; exit 1)
