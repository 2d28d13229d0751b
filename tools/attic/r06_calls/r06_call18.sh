mkdir -p gpurun_out
( for rep in 1 2; do for a in 0 1 2 4 8 16 31; do EFTS_LIB=$GRAFT_REPO_ROOT/lab/fft_abl$a.so timeout 100 python tools/gpu_probe_logmel_fft.py 2>&1 | grep "launch"; done; done ) > gpurun_out/fft_abl.txt; cat gpurun_out/fft_abl.txt
