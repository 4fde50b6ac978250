export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -s KILL 900 python -m pytest tests/test_gpu_gemm_pp.py -m gpu -q -x --timeout 300 > gpurun_out/pytest_pp.log 2>&1
echo "pp tests exit $?"; tail -8 gpurun_out/pytest_pp.log
timeout 600 python tools/experiments/gemm_pp_bench.py --variants 0,b0 --only qkv_plain --qkv --rounds 4 --reps 30 > gpurun_out/r06_pp_vt1.jsonl 2> gpurun_out/r06_pp_vt1.err
echo "bench rc $?"; tail -3 gpurun_out/r06_pp_vt1.err; cat gpurun_out/r06_pp_vt1.jsonl
