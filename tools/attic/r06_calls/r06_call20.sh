mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_frontend.py tests/test_cli_data.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert|radix" | tail -12 ) > gpurun_out/t_fe.txt; cat gpurun_out/t_fe.txt
( for rep in 1 2; do timeout 100 python tools/gpu_probe_logmel_fft.py 2>&1 | grep "launch"; done ) > gpurun_out/fft_probe.txt; cat gpurun_out/fft_probe.txt
for r in 0 1 2 4 8; do EFTS_LOGMEL_RADIX=$r timeout 120 python bench.py --workload logmel64 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('radix $r', round(d['ms_per_step'],4), 'ms per call;', round(r['avg_launch_us'],1), 'us', 'call' if $r == 0 else 'product', ';', r['bound'], round(r['frac'],3), 'of the roof')"; done > gpurun_out/logmel_radix_r06.txt 2>&1; cat gpurun_out/logmel_radix_r06.txt
NOPMC=1 WL=logmel64 STEPS=5 TSTEPS=20 TWARM=5 TIMELINE=12 bash tools/prof_conv.sh r06_logmel > /dev/null 2>&1; head -8 gpurun_out/prof_r06_logmel/summary_r06_logmel.txt | cut -c1-150
timeout 200 python bench.py --workload logmel64 > gpurun_out/bench_logmel64_r06.json 2>/dev/null; cut -c1-600 gpurun_out/bench_logmel64_r06.json
