set -e
mkdir -p gpurun_out/r03g
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_e2e.py -x -q -m gpu -k "gradient or train or ss2d_core or backward or block or e2e" > gpurun_out/r03g/pytest.txt 2>&1 || { tail -40 gpurun_out/r03g/pytest.txt; exit 1; }
tail -2 gpurun_out/r03g/pytest.txt
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-latency --no-cpu-baseline > gpurun_out/r03g/bench.json 2> gpurun_out/r03g/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03g/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); t=d['train']; print({k:t[k] for k in ('value','ms_per_step','launch','eager','graphed') if k in t})
PY
