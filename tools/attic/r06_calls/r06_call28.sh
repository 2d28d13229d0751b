mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_align_gpu.py tests/test_gpu_parity.py -m gpu -x -q -k "prenet or frame_linear or forward" 2>&1 | grep -E "passed|failed|rror|assert" | tail -4 ) > gpurun_out/t_pn.txt; cat gpurun_out/t_pn.txt
for rep in 1 2 3; do for v in 16 8; do if [ $v = 8 ]; then export EFTS_AB_FL8=1; else unset EFTS_AB_FL8; fi; SHAPE=64,128,800 timeout 300 python tools/gpu_ab_forward.py base= 2>&1 | grep forward | sed "s/base/waves $v/"; done; done > gpurun_out/ab_prenet_waves.txt; cat gpurun_out/ab_prenet_waves.txt
