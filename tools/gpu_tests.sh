#!/bin/bash
# GPU-box driver: parity tests with hard timeouts (a hung kernel must not eat the box).
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > gpurun_out/device.txt
timeout -s KILL ${TEST_TIMEOUT:-900} python -m pytest tests -m gpu -q -x --timeout 300 ${PYTEST_ARGS} > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -40 gpurun_out/pytest_gpu.log
