cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu --timeout 120 -k "stream_k_matches_the_plain or merge_parts" > gpurun_out/r05_t15.log 2>&1; tail -4 gpurun_out/r05_t15.log
S=8394x3072x768,8394x768x3072,8394x768x768
for act in 0 1; do
timeout 300 python tools/experiments/gemm_variant_bench.py run --act $act --variants base --shapes $S 2>&1 | grep variant
GEMM_SK=1 timeout 300 python tools/experiments/gemm_variant_bench.py run --act $act --variants base --shapes $S 2>&1 | grep variant
done
