set -e
mkdir -p gpurun_out/r02m
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02m/smoke.txt 2>&1 || (tail -20 gpurun_out/r02m/smoke.txt; exit 1)
tail -1 gpurun_out/r02m/smoke.txt
( time python bench.py ) > gpurun_out/r02m/bench_default.json 2> gpurun_out/r02m/bench_default.err
grep real gpurun_out/r02m/bench_default.err
python bench.py --steps 200 --warmup 50 --cpu-baseline-full > gpurun_out/r02m/bench_long.json 2> gpurun_out/r02m/bench_long.err
