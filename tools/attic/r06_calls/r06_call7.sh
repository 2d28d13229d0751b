mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_frontend.py -m gpu -x -q 2>&1 | grep -v "Warn\|warn\|^$" | tail -12 ) > gpurun_out/t_frontend.txt; tail -4 gpurun_out/t_frontend.txt
for r in 1 2 4 8; do EFTS_LOGMEL_RADIX=$r timeout 120 python bench.py --workload logmel64 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('radix $r', round(d['ms_per_step'],4), 'ms', d['roofline']['avg_launch_us'], 'us product', d['roofline']['pipeline'])"; done > gpurun_out/logmel_radix_r06.txt 2>&1; cat gpurun_out/logmel_radix_r06.txt
SKIP_LAB=1 timeout 2400 bash tools/collect_r06.sh > gpurun_out/collect_r06.log 2>&1; tail -5 gpurun_out/collect_r06.log
