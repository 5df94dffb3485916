set -e
mkdir -p gpurun_out/r03q
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu -k "dwconv or dw or gradient or block or train" > gpurun_out/r03q/pytest.txt 2>&1 || { tail -40 gpurun_out/r03q/pytest.txt; exit 1; }
tail -2 gpurun_out/r03q/pytest.txt
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-latency --no-cpu-baseline > gpurun_out/r03q/bench.json 2> gpurun_out/r03q/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03q/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); t=d['train']; print({k:t[k] for k in ('value','ms_per_step','launch','eager','graphed') if k in t})
PY
