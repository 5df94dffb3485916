set -e
mkdir -p gpurun_out/r02a
python scripts/measure_lowp_parity.py > gpurun_out/r02a/lowp.json 2> gpurun_out/r02a/lowp.err
python scripts/bench_scan.py helix0 enc0 helix1 helix2 > gpurun_out/r02a/bench_scan.txt 2>&1
python bench.py --steps 30 --warmup 10 > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err
