cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 200 python tools/experiments/shift_timeline.py run --md gpurun_out/r05_shift_timeline_c.md > gpurun_out/r05_shift_timeline_c.log 2>&1; tail -8 gpurun_out/r05_shift_timeline_c.log
for q in 4 8; do for d in 1 0; do
GPU_MAX_HW_QUEUES=$q AS_DEFER_FPN=$d AS_BENCH_OTHER_RNG=0 AS_BENCH_MIL=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --other-configs "" > gpurun_out/r05_bench_q${q}_d$d.json 2> gpurun_out/r05_bench_q${q}_d$d.err
echo "queues $q defer $d: $(cut -c100-180 gpurun_out/r05_bench_q${q}_d$d.json)"
done; done
