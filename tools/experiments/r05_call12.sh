cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
AS_BENCH_EVENTS=0 AS_BENCH_OTHER_RNG=0 AS_BENCH_MIL=0 PROF_LINES=70 tools/prof_cmd.sh r05_bench_kernel_stats_mid python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --other-configs "" > /dev/null 2>&1
(cd _r04 && AS_BENCH_EVENTS=0 AS_BENCH_OTHER_RNG=0 AS_BENCH_MIL=0 PROF_LINES=70 tools/prof_cmd.sh r04tree_bench_kernel_stats python $GRAFT_REPO_ROOT/_r04/bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --other-configs "" > /dev/null 2>&1; cp gpurun_out/*r04tree* $GRAFT_REPO_ROOT/gpurun_out/ 2>/dev/null)
ls gpurun_out | grep -i "kernel_stats" | tail -5
