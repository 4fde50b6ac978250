export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/experiments/gemm_pp_bench.py --lib --qkv --rounds 5 --reps 30 > gpurun_out/r06_pp_bench2.jsonl 2> gpurun_out/r06_pp_bench2.err
echo "bench rc $?"
tail -5 gpurun_out/r06_pp_bench2.err
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r06_pp_bench2.jsonl")]
for r in rows:
    if "variant" in r:
        print(f'{r["shape"]:10s} {r["variant"]:4s} {r["us_min"]:7.1f} us {r["tflops"]:5d} TF err {r.get("err")} nan {r.get("nan")} vs_old {r.get("vs_old_max")} repro {r.get("bitwise_repro")}')
    else:
        print(r)
PY
