export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -s KILL 900 python -m pytest tests/test_gpu_gemm_pp.py -m gpu -q -x --timeout 600 > gpurun_out/pytest_pp.log 2>&1
echo "pp tests exit $?"; tail -15 gpurun_out/pytest_pp.log
timeout 600 python tools/experiments/gemm_pp_ablate.py run --variants nodefer,base,fill4,fill9 --cfgs b0 --shapes fc1,fc1_gelu,qkv,fc2,proj --rounds 4 --reps 30 > gpurun_out/r06_pp_defer1.jsonl 2> gpurun_out/r06_pp_defer1.err
echo "ablate rc $?"; tail -3 gpurun_out/r06_pp_defer1.err
timeout 300 python tools/experiments/gemm_pp_bench.py --variants 0,a0,b0 --qkv --rounds 3 --reps 20 > gpurun_out/r06_pp_bench5.jsonl 2> gpurun_out/r06_pp_bench5.err
echo "bench rc $?"; tail -3 gpurun_out/r06_pp_bench5.err
python - <<'PY'
import json
for l in open("gpurun_out/r06_pp_defer1.jsonl"):
    r = json.loads(l)
    print(f'  {r["shape"]:8s} {r["variant"]:15s} {r["cfg"]:3s} {r["us_min"]:7.1f} us  {r["tflops"]:5d} TF')
for l in open("gpurun_out/r06_pp_bench5.jsonl"):
    r = json.loads(l)
    if "variant" in r:
        print(f'{r["shape"]:10s} {r["variant"]:4s} {r["us_min"]:7.1f} us {r["tflops"]:5d} TF err {r.get("err")} nan {r.get("nan")} vs_old {r.get("vs_old_max")} repro {r.get("bitwise_repro")}')
    else:
        print(r)
PY
