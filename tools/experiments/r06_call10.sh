export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -s KILL 900 python -m pytest tests/test_gpu_gemm_pp.py -m gpu -q -x --timeout 600 > gpurun_out/pytest_pp.log 2>&1
echo "pp tests exit $?"; tail -5 gpurun_out/pytest_pp.log
timeout 800 python tools/experiments/gemm_pp_ablate.py run --variants nodefer,base,prio0,prio2,nost --cfgs b0 --shapes fc1,fc1_gelu,qkv,fc2,proj --rounds 4 --reps 30 > gpurun_out/r06_pp_defer6.jsonl 2> gpurun_out/r06_pp_defer6.err
echo "ablate rc $?"; tail -3 gpurun_out/r06_pp_defer6.err
python - <<'PY'
import json
for l in open("gpurun_out/r06_pp_defer6.jsonl"):
    r = json.loads(l)
    print(f'  {r["shape"]:8s} {r["variant"]:15s} {r["cfg"]:3s} {r["us_min"]:7.1f} us  {r["tflops"]:5d} TF')
PY
