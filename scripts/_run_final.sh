set -e
TAG=r02f
mkdir -p gpurun_out/$TAG
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/$TAG/pytest.txt 2>&1 || { tail -40 gpurun_out/$TAG/pytest.txt; exit 1; }
tail -2 gpurun_out/$TAG/pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
echo tests-done
