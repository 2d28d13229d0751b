mkdir -p gpurun_out
timeout 900 python tools/gpu_ab_forward.py base= early=defer_dur_tail:0 2>&1 | grep forward > gpurun_out/ab_defer.txt; cat gpurun_out/ab_defer.txt
