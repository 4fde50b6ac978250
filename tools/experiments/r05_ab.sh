# same-box A/B of the headline leg: the round-4 tree (git archive b73500e into _r04/, built there) against this tree
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for i in 1 2 3; do
  (cd _r04 && AS_BENCH_OTHER_RNG=0 AS_BENCH_MIL=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --other-configs "" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r04', d['value'], d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline_affinity']['ms_per_call'])")
  AS_BENCH_OTHER_RNG=0 AS_BENCH_MIL=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --other-configs "" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r05', d['value'], d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline_affinity']['ms_per_call'], d.get('roofline_attention_block',{}).get('ms_per_layer'))"
done
