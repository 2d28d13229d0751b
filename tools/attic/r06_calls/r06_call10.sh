mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_frontend.py -m gpu -x -q 2>&1 | grep -v "Warn\|warn\|^$" | tail -12 ) > gpurun_out/t_fe.txt; tail -3 gpurun_out/t_fe.txt
for r in 1 2 4 8; do EFTS_LOGMEL_RADIX=$r timeout 120 python bench.py --workload logmel64 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('radix $r', round(d['ms_per_step'],4), 'ms per call;', round(d['roofline']['avg_launch_us'],1), 'us product;', 'pipeline', round(d['roofline']['pipeline']['call_us'],1), 'us,', round(d['roofline']['pipeline']['frac'],3), 'of HBM')"; done > gpurun_out/logmel_radix_r06.txt 2>&1; cat gpurun_out/logmel_radix_r06.txt
NOPMC=1 WL=logmel64 STEPS=5 TSTEPS=20 TWARM=5 TIMELINE=12 bash tools/prof_conv.sh r06_logmel > /dev/null 2>&1; head -8 gpurun_out/prof_r06_logmel/summary_r06_logmel.txt | cut -c1-150
timeout 200 python bench.py --workload logmel64 > gpurun_out/bench_logmel64_r06.json 2>/dev/null
