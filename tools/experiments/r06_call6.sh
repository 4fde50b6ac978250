export TMPDIR=/tmp
mkdir -p gpurun_out
t0=$(date +%s)
timeout -s KILL 1500 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $? after $(( $(date +%s) - t0 )) s" | tee -a gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r06_bench_a.json 2> gpurun_out/r06_bench_a.log
echo "bench rc $?"; tail -3 gpurun_out/r06_bench_a.log
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r06_bench_a.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "images_per_sec_reference_rng", "images_per_sec_fp32_parity_path", "images_per_sec_mil_selector"):
    print(k, r.get(k))
print("roofline", r.get("roofline"))
print("block", r.get("roofline_attention_block"))
print("train", {k: v for k, v in (r.get("train") or {}).items() if k in ("ms_per_step", "images_per_sec", "error")})
PY
