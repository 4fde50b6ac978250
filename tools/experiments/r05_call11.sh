cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -x -m gpu --timeout 300 > gpurun_out/r05_t11.log 2>&1; tail -5 gpurun_out/r05_t11.log
for i in 1 2; do for g in 0 1; do
AS_HEAD_TENSOR_GLUE=$g AS_BENCH_OTHER_RNG=0 AS_BENCH_MIL=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --other-configs "" > gpurun_out/r05_bench11_$g.json 2> gpurun_out/r05_bench11_$g.err
echo "tensor glue $g: $(cut -c100-200 gpurun_out/r05_bench11_$g.json)"
done; done
timeout 300 python tools/experiments/glue_sites.py --steps 3 --rows 10 2>&1 | grep "ATen ops on device\|ATen device time" 
