cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -q -x -m gpu --timeout 200 -k "shift or cluster or mean_shift" > gpurun_out/r05_t5.log 2>&1; tail -3 gpurun_out/r05_t5.log
timeout 200 python tools/experiments/shift_timeline.py run --md gpurun_out/r05_shift_timeline_b.md > gpurun_out/r05_shift_timeline_b.log 2>&1; tail -8 gpurun_out/r05_shift_timeline_b.log
timeout 300 python tools/experiments/step_trace.py > gpurun_out/r05_step_scopes_defer.log 2>&1
cp gpurun_out/step_timeline.txt gpurun_out/r05_step_timeline_defer.txt
rm -f gpurun_out/step_trace.json
tail -3 gpurun_out/r05_step_scopes_defer.log
