export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 400 python tools/experiments/gemm_pp_bench.py --lib --qkv --rounds 5 --reps 30 > gpurun_out/r06_pp_bench1.jsonl 2> gpurun_out/r06_pp_bench1.err
echo "bench rc $?"
tail -5 gpurun_out/r06_pp_bench1.err
cat gpurun_out/r06_pp_bench1.jsonl
PROF_NAME_WIDTH=400 PROF_LINES=30 timeout 300 tools/prof_cmd.sh r06_hipblaslt_solutions python $GRAFT_REPO_ROOT/tools/experiments/hipblaslt_names.py
