cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for m in half 3of4 1of4; do for q in 4 8; do
GPU_MAX_HW_QUEUES=$q AS_FPN_CU_MASK=$m AS_DEFER_FPN=1 AS_BENCH_OTHER_RNG=0 AS_BENCH_MIL=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --other-configs "" > gpurun_out/r05_bench_m${m}_q$q.json 2> gpurun_out/r05_bench_m${m}_q$q.err
echo "mask $m queues $q: $(cut -c100-180 gpurun_out/r05_bench_m${m}_q$q.json) $(tail -1 gpurun_out/r05_bench_m${m}_q$q.err | cut -c1-200)"
done; done
GPU_MAX_HW_QUEUES=4 AS_DEFER_FPN=0 AS_BENCH_OTHER_RNG=0 AS_BENCH_MIL=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --other-configs "" > gpurun_out/r05_bench_ref.json 2>/dev/null
echo "ref: $(cut -c100-180 gpurun_out/r05_bench_ref.json)"
