set -e
mkdir -p gpurun_out/r02b
python scripts/bench_scan.py helix0 helix1 helix2 enc2 > gpurun_out/r02b/bench_scan.txt 2>&1
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_properties.py tests/test_gpu_model.py -x -q > gpurun_out/r02b/pytest.txt 2>&1 || (tail -30 gpurun_out/r02b/pytest.txt; exit 1)
python scripts/measure_lowp_parity.py > gpurun_out/r02b/lowp.json 2> gpurun_out/r02b/lowp.err
