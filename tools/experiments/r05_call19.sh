cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for i in 1 2; do
for cfg in "0 1" "1 1" "1 2" "1 0"; do set -- $cfg
AS_DEFER_FPN=$1 AS_FPN_TILE_HINT=$2 AS_BENCH_EVENTS=0 AS_BENCH_OTHER_RNG=0 AS_BENCH_MIL=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --other-configs "" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('defer $1 hint $2', d['value'], d['ms_per_step'])"
done; done
