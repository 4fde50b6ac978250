cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for i in 1 2 3; do for q in 4 8 6; do
GPU_MAX_HW_QUEUES=$q AS_BENCH_EVENTS=0 AS_BENCH_OTHER_RNG=0 AS_BENCH_MIL=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --other-configs "" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $q', d['value'], d['ms_per_step'])"
done; done
