mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_frontend.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -4 ) > gpurun_out/t_fe.txt; cat gpurun_out/t_fe.txt
( for rep in 1 2 3; do EFTS_AB_FFT4=1 timeout 100 python tools/gpu_probe_logmel_fft.py 2>&1 | grep "launch" | sed 's/product/4-wave x3/'; timeout 100 python tools/gpu_probe_logmel_fft.py 2>&1 | grep "launch" | sed 's/product/16-wave x1/'; done ) > gpurun_out/fft_probe_ab.txt; cat gpurun_out/fft_probe_ab.txt
