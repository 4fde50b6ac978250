cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_path.py -q -x -m gpu --timeout 300 -k "merge_parts or seed_pseudo or semantic_post or fast_mode" > gpurun_out/r05_t14.log 2>&1; tail -3 gpurun_out/r05_t14.log
bash tools/experiments/r05_ab.sh 2>&1 | tee gpurun_out/r05_ab3.log
AS_BENCH_EVENTS=0 AS_BENCH_OTHER_RNG=0 AS_BENCH_MIL=0 PROF_LINES=2 tools/prof_cmd.sh r05_bench_kernel_stats_mid3 python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --other-configs "" > /dev/null 2>&1
grep "merge_parts\|total kernel" gpurun_out/r05_bench_kernel_stats_mid3.md | cut -c1-200
