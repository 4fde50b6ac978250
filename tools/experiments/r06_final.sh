export TMPDIR=/tmp
mkdir -p gpurun_out
t0=$(date +%s)
timeout -s KILL 1800 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $? after $(( $(date +%s) - t0 )) s" | tee -a gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/round_artifacts.sh r06 2>&1 | tail -3
