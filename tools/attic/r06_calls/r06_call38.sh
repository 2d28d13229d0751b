mkdir -p gpurun_out
WL=logmel64 STEPS=5 TSTEPS=20 TWARM=5 TIMELINE=12 bash tools/prof_conv.sh r06_logmel > /dev/null 2>&1
grep -n "logmel_fft" gpurun_out/prof_r06_logmel/summary_r06_logmel.txt | cut -c1-170 | head -40
