cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -x -m gpu --timeout 300 > gpurun_out/r05_t16.log 2>&1; tail -4 gpurun_out/r05_t16.log
bash tools/experiments/r05_ab.sh 2>&1 | tee gpurun_out/r05_ab4.log
AS_BENCH_EVENTS=0 AS_BENCH_OTHER_RNG=0 AS_BENCH_MIL=0 PROF_LINES=2 tools/prof_cmd.sh r05_bench_kernel_stats_mid4 python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --other-configs "" > /dev/null 2>&1
grep "merge_parts\|total kernel" gpurun_out/r05_bench_kernel_stats_mid4.md | cut -c1-200
