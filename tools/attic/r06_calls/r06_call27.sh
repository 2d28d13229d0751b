# final training evidence at HEAD (pack_vt64) + vocoder8 line + the driver's own command
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/collect; mkdir -p $O; cd $R
WL=train32 STEPS=3 TSTEPS=3 TWARM=2 TIMELINE=240 BARGS="--train-graph 0" bash tools/prof_conv.sh r06_train_bf16 > /dev/null 2>&1
EFTS_BENCH_TRAIN_NO_EAGER=1 WL=train32 STEPS=4 TSTEPS=20 TWARM=5 TIMELINE=215 BARGS="--train-graph 1" bash tools/prof_conv.sh r06_train_graph > /dev/null 2>&1
for t in r06_train_bf16 r06_train_graph; do test -s gpurun_out/prof_$t/summary_$t.txt; cp gpurun_out/prof_$t/summary_$t.txt $O/rocprofv3_${t}_summary.txt; done
python bench.py --workload train32 > $O/bench_train32_bf16_r06.json 2> $O/bench_train32.err
python bench.py --workload train32 --precision bf16x3 --no-cpu-baseline > $O/bench_train32_bf16x3_r06.json 2>> $O/bench_train32.err
python bench.py --workload vocoder8 --no-cpu-baseline > $O/bench_vocoder8_r06.json 2>> $O/bench_train32.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_fwd64_driverargs_r06.json 2>> $O/bench_train32.err
ls -la $O | tail -8
