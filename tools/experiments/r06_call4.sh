export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/experiments/gemm_pp_ablate.py run --variants prev,base,prio0,prio2,abl3,abl1,abl2 --cfgs a0,b0 --shapes sq4096,fc1,fc2 --rounds 4 --reps 30 > gpurun_out/r06_pp_ablate2.jsonl 2> gpurun_out/r06_pp_ablate2.err
echo "ablate rc $?"; tail -3 gpurun_out/r06_pp_ablate2.err
python - <<'PY'
import json
for l in open("gpurun_out/r06_pp_ablate2.jsonl"):
    r = json.loads(l)
    print(f'  {r["shape"]:8s} {r["variant"]:5s} {r["cfg"]:3s} {r["us_min"]:7.1f} us  {r["tflops"]:5d} TF')
for l in open("gpurun_out/r06_pp_bench3.jsonl"):
    r = json.loads(l)
    if "variant" in r:
        print(f'{r["shape"]:10s} {r["variant"]:4s} {r["us_min"]:7.1f} us {r["tflops"]:5d} TF err {r.get("err")} nan {r.get("nan")} vs_old {r.get("vs_old_max")} repro {r.get("bitwise_repro")}')
    else:
        print(r)
PY
