export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/experiments/gemm_pp_ablate.py run --variants base,abl1,abl2,abl3,abl6 --cfgs a0,b0 --shapes sq4096,fc1 --rounds 4 --reps 30 > gpurun_out/r06_pp_ablate1.jsonl 2> gpurun_out/r06_pp_ablate1.err
echo "ablate rc $?"; tail -3 gpurun_out/r06_pp_ablate1.err
timeout 300 python tools/experiments/gemm_pp_ablate.py run --variants base --cfgs a0,a2,b0,b2 --shapes sq4096,fc1,fc2,proj,qkv --rounds 4 --reps 30 > gpurun_out/r06_pp_var2.jsonl 2> gpurun_out/r06_pp_var2.err
echo "var2 rc $?"; tail -3 gpurun_out/r06_pp_var2.err
python - <<'PY'
import json
for f in ("gpurun_out/r06_pp_ablate1.jsonl", "gpurun_out/r06_pp_var2.jsonl"):
    print(f)
    for l in open(f):
        r = json.loads(l)
        print(f'  {r["shape"]:8s} {r["variant"]:5s} {r["cfg"]:3s} {r["us_min"]:7.1f} us  {r["tflops"]:5d} TF')
PY
