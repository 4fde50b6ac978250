export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 800 python tools/experiments/gemm_pp_ablate.py run --variants nodefer,base,vmx1,mode2,nost,nt,sc1 --cfgs b0 --shapes fc1,fc1_gelu,qkv,fc2 --rounds 4 --reps 30 > gpurun_out/r06_pp_defer2.jsonl 2> gpurun_out/r06_pp_defer2.err
echo "ablate rc $?"; tail -3 gpurun_out/r06_pp_defer2.err
python - <<'PY'
import json
for l in open("gpurun_out/r06_pp_defer2.jsonl"):
    r = json.loads(l)
    print(f'  {r["shape"]:8s} {r["variant"]:15s} {r["cfg"]:3s} {r["us_min"]:7.1f} us  {r["tflops"]:5d} TF')
PY
