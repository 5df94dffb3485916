set -e
mkdir -p gpurun_out/r03i
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu -k "layernorm or norm or linear or block or gradient or train" > gpurun_out/r03i/pytest.txt 2>&1 || { tail -40 gpurun_out/r03i/pytest.txt; exit 1; }
tail -2 gpurun_out/r03i/pytest.txt
timeout -k 10 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/r03i/bench.json 2> gpurun_out/r03i/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03i/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['latency_b1']['mean_ms']); t=d['train']; print({k:t[k] for k in ('value','ms_per_step','launch','eager','graphed') if k in t})
PY
