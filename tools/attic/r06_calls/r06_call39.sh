mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_frontend.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -4 ) > gpurun_out/t_fe.txt; cat gpurun_out/t_fe.txt
( for rep in 1 2; do timeout 100 python tools/gpu_probe_logmel_fft.py 2>&1 | grep "launch"; done ) > gpurun_out/fft_probe.txt; cat gpurun_out/fft_probe.txt
WL=logmel64 STEPS=5 TSTEPS=20 TWARM=5 TIMELINE=12 bash tools/prof_conv.sh r06_logmel > /dev/null 2>&1
grep -n "logmel_fft" gpurun_out/prof_r06_logmel/summary_r06_logmel.txt | grep -E "LDS|INSTS_VALU|WAIT_ANY|WAVE_CYCLES|avg=" | cut -c1-170
