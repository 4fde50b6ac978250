cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_path.py -q -x -m gpu --timeout 200 -k "deferred or backbone or seed_pseudo" > gpurun_out/r05_t4.log 2>&1; tail -3 gpurun_out/r05_t4.log
timeout 200 python tools/experiments/shift_timeline.py run --md gpurun_out/r05_shift_timeline.md > gpurun_out/r05_shift_timeline.log 2>&1; tail -5 gpurun_out/r05_shift_timeline.log
for d in 1 0; do
AS_DEFER_FPN=$d AS_BENCH_OTHER_RNG=0 AS_BENCH_MIL=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --other-configs "" > gpurun_out/r05_bench_defer$d.json 2> gpurun_out/r05_bench_defer$d.err
cut -c1-330 gpurun_out/r05_bench_defer$d.json
done
