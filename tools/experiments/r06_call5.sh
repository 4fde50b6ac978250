export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/experiments/gemm_pp_ablate.py run --variants glds,base,dma,dma_noswz,dma_glds,dma_glds_noswz,nomfma --cfgs a0,b0 --shapes sq4096,fc1,fc2 --rounds 4 --reps 30 > gpurun_out/r06_pp_ablate3.jsonl 2> gpurun_out/r06_pp_ablate3.err
echo "ablate rc $?"; tail -3 gpurun_out/r06_pp_ablate3.err
timeout 300 python tools/experiments/gemm_pp_bench.py --variants 0,a0,b0 --qkv --rounds 3 --reps 20 > gpurun_out/r06_pp_bench4.jsonl 2> gpurun_out/r06_pp_bench4.err
echo "bench rc $?"; tail -3 gpurun_out/r06_pp_bench4.err
python - <<'PY'
import json
for l in open("gpurun_out/r06_pp_ablate3.jsonl"):
    r = json.loads(l)
    print(f'  {r["shape"]:8s} {r["variant"]:15s} {r["cfg"]:3s} {r["us_min"]:7.1f} us  {r["tflops"]:5d} TF')
for l in open("gpurun_out/r06_pp_bench4.jsonl"):
    r = json.loads(l)
    if "variant" in r:
        print(f'{r["shape"]:10s} {r["variant"]:4s} {r["us_min"]:7.1f} us {r["tflops"]:5d} TF err {r.get("err")} nan {r.get("nan")} vs_old {r.get("vs_old_max")} repro {r.get("bitwise_repro")}')
    else:
        print(r)
PY
