export TMPDIR=/tmp
mkdir -p gpurun_out
R=r06
AS_BENCH_EVENTS=0 AS_BENCH_OTHER_RNG=0 AS_BENCH_MIL=0 AS_BENCH_FP32=0 PROF_LINES=8 tools/prof_cmd.sh ${R}_bench_kernel_stats_final python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --other-configs "" > /dev/null 2>&1
head -30 gpurun_out/${R}_bench_kernel_stats_final.md | cut -c1-150
timeout 300 python tools/experiments/step_trace.py > gpurun_out/${R}_step_trace.log 2>&1 && cp gpurun_out/step_timeline.txt gpurun_out/${R}_step_timeline.txt
timeout 300 python tools/experiments/glue_sites.py --steps 3 --rows 200 > gpurun_out/${R}_glue_sites.log 2>&1
tail -5 gpurun_out/${R}_glue_sites.log
