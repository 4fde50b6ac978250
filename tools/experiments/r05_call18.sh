cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python bench.py > gpurun_out/r05_bench_full.json 2> gpurun_out/r05_bench_full.err
echo "wall $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_bench_full.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','images_per_sec_reference_rng','images_per_sec_mil_selector') if k in d})
print('roofline',d['roofline']['frac'],d['roofline']['ms_per_launch'])
print('block',d.get('roofline_attention_block',{}).get('frac'),d.get('roofline_attention_block',{}).get('ms_per_layer'))
print('affinity',d['roofline_affinity']['ms_per_call'],d['roofline_affinity']['frac'])
print('cpu',d.get('cpu_baseline'))
t=d.get('train',{})
print('train',{k:t.get(k) for k in ('ms_per_step','images_per_sec','ms_per_step_accum2')}, str(t)[:600])
print('other',json.dumps(d.get('other_configs'))[:800])
PY
tail -3 gpurun_out/r05_bench_full.err
