set -e
mkdir -p gpurun_out/r02q
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "scan_forms" > gpurun_out/r02q/pytest.txt 2>&1 || (tail -30 gpurun_out/r02q/pytest.txt; exit 1)
tail -2 gpurun_out/r02q/pytest.txt
cd scripts && timeout -k 10 300 python bench_scan_w.py enc2 enc1 helix1 2>&1 | grep -v amdgpu | grep "form 0\|W 4\|W 8"
