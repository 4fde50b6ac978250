cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -x -m gpu --timeout 300 > gpurun_out/r05_gpu_tests_final.log 2>&1; tail -3 gpurun_out/r05_gpu_tests_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | tail -2
t0=$(date +%s); timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_style.json 2> gpurun_out/r05_bench_driver_style.err; echo "bench wall $(( $(date +%s) - t0 )) s rc $?"
cut -c1-300 gpurun_out/r05_bench_driver_style.json
