mkdir -p gpurun_out
( for rep in 1 2; do EFTS_LIB=$GRAFT_REPO_ROOT/lab/fft_old.so timeout 100 python tools/gpu_probe_logmel_fft.py 2>&1 | grep "launch"; timeout 100 python tools/gpu_probe_logmel_fft.py 2>&1 | grep "launch"; done ) > gpurun_out/fft_probe_ab.txt; cat gpurun_out/fft_probe_ab.txt
timeout 200 python bench.py --workload logmel64 > gpurun_out/bench_logmel64_r06.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/bench_logmel64_r06.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['cpu_baseline']['value'])"
WL=logmel64 STEPS=5 TSTEPS=20 TWARM=5 TIMELINE=12 bash tools/prof_conv.sh r06_logmel > /dev/null 2>&1
head -6 gpurun_out/prof_r06_logmel/summary_r06_logmel.txt | cut -c1-150
