set -e
mkdir -p gpurun_out/r02o
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_e2e.py -x -q -k "train or graphed or current_weights or disk" > gpurun_out/r02o/pytest.txt 2>&1 || (tail -40 gpurun_out/r02o/pytest.txt; exit 1)
tail -2 gpurun_out/r02o/pytest.txt
python bench.py --no-cpu-baseline --no-latency --steps 60 --warmup 20 > gpurun_out/r02o/bench.json 2> gpurun_out/r02o/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r02o/bench.json') if l.startswith('{')][-1])
t=d['train']; print('fwd', d['value'], 'train', t['value'], t['ms_per_step'], t['graphed'].get('ms_per_step'))
PY
