mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_frontend.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert|radix" | tail -12 ) > gpurun_out/t_fe.txt; cat gpurun_out/t_fe.txt
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mel_loss" 2>&1 | grep -E "passed|failed|rror|assert" | tail -5 ) > gpurun_out/t_loss.txt; cat gpurun_out/t_loss.txt
for r in 0 4 0 4; do EFTS_LOGMEL_RADIX=$r timeout 120 python bench.py --workload logmel64 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('radix $r', round(d['ms_per_step'],4), 'ms per call;', round(r['avg_launch_us'],1), 'us launch;', r['bound'], round(r['frac'],3))"; done > gpurun_out/logmel_fft_ab.txt 2>&1; cat gpurun_out/logmel_fft_ab.txt
timeout 600 python tools/gpu_ab_forward.py base= nofuse=fuse_mel_loss:0 2>&1 | grep forward > gpurun_out/ab_mel_loss.txt; cat gpurun_out/ab_mel_loss.txt
