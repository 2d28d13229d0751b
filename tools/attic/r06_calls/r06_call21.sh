mkdir -p gpurun_out
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5 ) > gpurun_out/t_gpu_all.txt; cat gpurun_out/t_gpu_all.txt
( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_fwd64_r06.json 2> gpurun_out/bench_fwd64.err ); python - <<'P'
import json
d=json.loads(open('gpurun_out/bench_fwd64_r06.json').read().strip().splitlines()[-1])
print('fwd64', d['ms_per_step'], d['value'], 'frac', d['roofline']['frac'], 'traffic', d['roofline'].get('traffic'))
for k in ('parity_mode','long16','infer64','infer_lj','train32'):
    v=d.get(k)
    if isinstance(v,dict): print(k, v.get('ms_per_step'), v.get('hip_vs_oracle_mel_max_abs'), (v.get('parity_mode') or {}).get('ms_per_step') if isinstance(v.get('parity_mode'),dict) else '')
print('stock', {k:(v.get('ms_per_step') if isinstance(v,dict) else v) for k,v in d.get('stock_gpu_baseline',{}).items() if k.startswith(('fwd','train'))})
print('cpu', d.get('cpu_baseline'))
P
