# same-box A/B of the headline leg: the previous commit's tree (git archive HEAD into _prev/, built there) against this tree,
# after the RoI-head and kernel GPU tests of this tree
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_path.py tests/test_gpu_kernels.py tests/test_gpu_cfg4_partb.py tests/test_gpu_fullsize.py -q -x -m gpu --timeout 300 2>&1 | tail -4
one() { AS_BENCH_OTHER_RNG=0 AS_BENCH_MIL=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --other-configs "" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['roofline_affinity']['ms_per_call'])"; }
for i in 1 2 3; do
  (cd _prev && one prev)
  one new
done
timeout 300 python tools/experiments/glue_sites.py > gpurun_out/r05_glue_sites.txt 2>&1; grep -n "ATen ops on device" gpurun_out/r05_glue_sites.txt
