set -e
mkdir -p gpurun_out/r02d
python -m pytest tests/test_gpu_kernels.py -x -q -k "wgrad or small_contractions" > gpurun_out/r02d/pytest_k.txt 2>&1 || (tail -40 gpurun_out/r02d/pytest_k.txt; exit 1)
python -m pytest tests/test_gpu_model.py -x -q -k "training_path or train_step or rccl or s_p_match" > gpurun_out/r02d/pytest_m.txt 2>&1 || (tail -60 gpurun_out/r02d/pytest_m.txt; exit 1)
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -x -q > gpurun_out/r02d/pytest_rest.txt 2>&1 || (tail -60 gpurun_out/r02d/pytest_rest.txt; exit 1)
python bench.py --no-cpu-baseline --no-latency --steps 20 > gpurun_out/r02d/bench.json 2> gpurun_out/r02d/bench.err
