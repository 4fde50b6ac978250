export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -s KILL 900 python -m pytest tests/test_gpu_gemm_pp.py -m gpu -q -x --timeout 300 > gpurun_out/pytest_pp.log 2>&1
echo "pp tests exit $?"; tail -8 gpurun_out/pytest_pp.log
timeout 600 python tools/experiments/gemm_pp_bench.py --variants 0,a0,a1,b0,b1 --qkv --rounds 4 --reps 20 > gpurun_out/r06_pp_sk1.jsonl 2> gpurun_out/r06_pp_sk1.err
echo "bench rc $?"; tail -3 gpurun_out/r06_pp_sk1.err
python - <<'PY'
import json
for l in open("gpurun_out/r06_pp_sk1.jsonl"):
    r = json.loads(l)
    if "variant" in r:
        print(f'{r["shape"]:10s} {r["variant"]:4s} {r["us_min"]:7.1f} us {r["tflops"]:5d} TF err {r.get("err")} nan {r.get("nan")} vs_old {r.get("vs_old_max")} repro {r.get("bitwise_repro")}')
    else:
        print(r)
PY
