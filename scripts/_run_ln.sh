set -e
mkdir -p gpurun_out/r02n
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "folded" > gpurun_out/r02n/pytest_k.txt 2>&1 || (tail -30 gpurun_out/r02n/pytest_k.txt; exit 1)
tail -2 gpurun_out/r02n/pytest_k.txt
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_properties.py tests/test_gpu_e2e.py -x -q > gpurun_out/r02n/pytest_m.txt 2>&1 || (tail -40 gpurun_out/r02n/pytest_m.txt; exit 1)
tail -2 gpurun_out/r02n/pytest_m.txt
python bench.py --no-cpu-baseline --steps 100 --warmup 30 --no-train > gpurun_out/r02n/bench.json 2> gpurun_out/r02n/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r02n/bench.json') if l.startswith('{')][-1])
print('fwd', d['value'], d['ms_per_step'], 'lat', d['latency_b1']['mean_ms'], 'gemm', d['roofline_gemm']['ms_per_step'], d['roofline_gemm']['launches'])
PY
python scripts/measure_lowp_parity.py > gpurun_out/r02n/lowp.json 2> gpurun_out/r02n/lowp.err
