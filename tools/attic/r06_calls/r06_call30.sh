mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_frontend.py tests/test_cli_data.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -4 ) > gpurun_out/t_fe.txt; cat gpurun_out/t_fe.txt
( for rep in 1 2; do timeout 100 python tools/gpu_probe_logmel_fft.py 2>&1 | grep "launch"; done ) > gpurun_out/fft_probe.txt; cat gpurun_out/fft_probe.txt
timeout 200 python bench.py --workload logmel64 > gpurun_out/bench_logmel64_r06.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/bench_logmel64_r06.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['cpu_baseline']['value'])"
