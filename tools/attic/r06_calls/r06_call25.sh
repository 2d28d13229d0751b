mkdir -p gpurun_out
NOPMC=1 EFTS_BENCH_TRAIN_NO_EAGER=1 WL=train32 STEPS=4 TSTEPS=20 TWARM=5 TIMELINE=215 BARGS="--train-graph 1" bash tools/prof_conv.sh r06b_train_graph > /dev/null 2>&1
grep -n "pack_vt" gpurun_out/prof_r06b_train_graph/summary_r06b_train_graph.txt | head -12
