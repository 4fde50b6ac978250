cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -x -m gpu --timeout 300 > gpurun_out/r05_t8.log 2>&1; tail -5 gpurun_out/r05_t8.log
timeout 200 python tools/experiments/gemm_variant_bench.py run --act 1 --variants base,base@tall --shapes 8394x3072x768 > gpurun_out/r05_gelu.log 2>&1; grep variant gpurun_out/r05_gelu.log
timeout 300 python tools/experiments/sdpa_impl_bench.py --variants 7,7@prio3,7@prio4 --rounds 7 --shapes 2x12x4197 > gpurun_out/r05_sdpa_prio.log 2>&1; tail -8 gpurun_out/r05_sdpa_prio.log
AS_BENCH_OTHER_RNG=0 AS_BENCH_MIL=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --other-configs "" > gpurun_out/r05_bench8.json 2> gpurun_out/r05_bench8.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_bench8.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('roofline_attention_block',{}).get('frac'), d.get('roofline_attention_block',{}).get('ms_per_layer'), d['roofline']['frac'], d['roofline_affinity']['ms_per_call'])
PY
