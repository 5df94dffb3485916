set -e
mkdir -p gpurun_out/r02g
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "scan_forms" > gpurun_out/r02g/pytest.txt 2>&1 || (tail -30 gpurun_out/r02g/pytest.txt; exit 1)
timeout -k 10 300 python scripts/bench_scan.py helix0 enc0 > gpurun_out/r02g/bench_scan.txt 2>&1
grep -v amdgpu.ids gpurun_out/r02g/bench_scan.txt
