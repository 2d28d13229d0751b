mkdir -p gpurun_out
timeout 100 python tools/gpu_probe_logmel_fft.py 2>&1 | grep "launch"
timeout 200 python bench.py --workload logmel64 > gpurun_out/bench_logmel64_r06.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/bench_logmel64_r06.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['cpu_baseline']['value'])"
NOPMC=1 WL=logmel64 STEPS=5 TSTEPS=20 TWARM=5 TIMELINE=12 bash tools/prof_conv.sh r06_logmel_fast > /dev/null 2>&1; head -5 gpurun_out/prof_r06_logmel_fast/summary_r06_logmel_fast.txt | tail -2 | cut -c1-120
