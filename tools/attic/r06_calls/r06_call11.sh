mkdir -p gpurun_out
NOPMC=1 PREC=bf16 TIMELINE=70 bash tools/prof_conv.sh r06b_bf16 > /dev/null 2>&1; grep -n "^span" gpurun_out/prof_r06b_bf16/summary_r06b_bf16.txt
NOPMC=1 PREC=bf16x3 TIMELINE=70 bash tools/prof_conv.sh r06b_bf16x3 > /dev/null 2>&1; grep -n "^span" gpurun_out/prof_r06b_bf16x3/summary_r06b_bf16x3.txt
( timeout 300 python tools/gpu_probe_fwd_skip.py bf16; timeout 300 python tools/gpu_probe_fwd_skip.py bf16x3 ) 2>&1 | grep BOUND > gpurun_out/fwd_skip_bounds_r06_before.txt; tail -14 gpurun_out/fwd_skip_bounds_r06_before.txt
