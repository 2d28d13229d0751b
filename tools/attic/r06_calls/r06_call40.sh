mkdir -p gpurun_out
( for rep in 1 2 3; do EFTS_LIB=$GRAFT_REPO_ROOT/lab/fft_old.so timeout 100 python tools/gpu_probe_logmel_fft.py 2>&1 | grep "launch"; timeout 100 python tools/gpu_probe_logmel_fft.py 2>&1 | grep "launch"; done ) > gpurun_out/fft_probe_ab.txt; cat gpurun_out/fft_probe_ab.txt
