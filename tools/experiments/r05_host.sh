cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_path.py tests/test_gpu_kernels.py -q -x -m gpu --timeout 300 2>&1 | tail -3
timeout 300 python tools/experiments/train_hostprof.py 70 > gpurun_out/r05_train_hostprof.txt 2>&1
timeout 300 python tools/experiments/train_phases.py 8 > gpurun_out/r05_train_phases_b.txt 2>&1; tail -9 gpurun_out/r05_train_phases_b.txt
