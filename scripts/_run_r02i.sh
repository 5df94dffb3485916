set -e
mkdir -p gpurun_out/r02i
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -k "training or train_step or gradients or scan_forms or epilogue or rccl" > gpurun_out/r02i/pytest.txt 2>&1 || (tail -40 gpurun_out/r02i/pytest.txt; exit 1)
tail -2 gpurun_out/r02i/pytest.txt
python bench.py --no-cpu-baseline --no-latency --steps 20 > gpurun_out/r02i/bench.json 2> gpurun_out/r02i/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r02i/bench.json') if l.startswith('{')][-1])
r=d['roofline']; t=d['train']
print('fwd', d['value'], 'pair', r['avg_us'], r['scan_us'], r['merge_us'], r['frac'], 'train', t['value'], t['ms_per_step'], t['graphed'])
PY
