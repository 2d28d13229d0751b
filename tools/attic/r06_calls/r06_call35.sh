mkdir -p gpurun_out
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_fwd64_driverargs_r06.json 2> gpurun_out/bench_fwd64.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/bench_fwd64_driverargs_r06.json').read().strip().splitlines()[-1]); t=d['train32']
print('fwd64', d['ms_per_step'], 'parity', d['parity_mode']['ms_per_step'], 'long16', d['long16']['ms_per_step'], 'infer64', d['infer64']['ms_per_step'], 'infer_lj', d['infer_lj']['ms_per_step'], d['infer_lj']['parity_mode']['ms_per_step'], 'train', t['ms_per_step'], t['parity_mode']['ms_per_step'])
P
