# same-box A/B: previous tree (_prev/) vs this tree with the point head in line / on its side stream
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
one() { AS_BENCH_OTHER_RNG=0 AS_BENCH_MIL=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --other-configs "" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['roofline_affinity']['ms_per_call'])"; }
for i in 1 2 3; do
  (cd _prev && one prev)
  AS_POINT_HEAD_STREAM=0 one new_inline
  AS_POINT_HEAD_STREAM=1 one new_side
done
