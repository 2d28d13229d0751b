mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_frontend.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert|radix" | tail -12 ) > gpurun_out/t_fe.txt; cat gpurun_out/t_fe.txt
( for rep in 1 2 3; do timeout 100 python tools/gpu_probe_logmel_fft.py 2>&1 | grep "launch"; done ) > gpurun_out/fft_abl.txt; cat gpurun_out/fft_abl.txt
