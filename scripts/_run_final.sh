set -e
TAG=r02g
bash scripts/collect_profiles.sh $TAG > /dev/null 2>&1
bash scripts/pmc_sq.sh $TAG > /dev/null 2>&1
python bench.py --steps 200 --warmup 50 > gpurun_out/$TAG/bench_long.json 2> gpurun_out/$TAG/bench_long.err
echo collected
