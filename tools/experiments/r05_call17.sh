cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_path.py -q -x -m gpu --timeout 300 -k "qkv_layout or attention_bf16 or attention_f32 or backbone or precision_matched" > gpurun_out/r05_t17.log 2>&1; tail -4 gpurun_out/r05_t17.log
timeout 200 python tools/experiments/qkv_ab.py base vtdirect 2>&1 | tail -8
