export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -s KILL 600 python -m pytest tests/test_gpu_backbone_fullsize.py -m gpu -q -x --timeout 500 -s > gpurun_out/pytest_fullsize.log 2>&1
echo "fullsize exit $?"; grep -E "bf16|fp32|passed|failed|Error|assert" gpurun_out/pytest_fullsize.log | tail -12
t0=$(date +%s)
timeout -s KILL 1800 python -m pytest tests -m gpu -q --timeout 600 --deselect tests/test_gpu_backbone_fullsize.py > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $? after $(( $(date +%s) - t0 )) s" | tee -a gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
