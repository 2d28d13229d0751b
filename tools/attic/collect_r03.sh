#!/bin/bash
# Round-3 evidence in one gpurun call: rocprofv3 summaries (trace + PMC passes), the bench lines of every workload and the
# workgroup stamps of the dominant kernel.  Everything lands in gpurun_out/collect/; copy what is to be judged into profiles/.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/collect
mkdir -p $O
cd $R
PREC=bf16 TIMELINE=60 bash tools/prof_conv.sh r03_bf16 > /dev/null 2>&1
PREC=bf16x3 bash tools/prof_conv.sh r03_bf16x3 > /dev/null 2>&1
WL=train32 STEPS=3 TIMELINE=240 bash tools/prof_conv.sh r03_train_bf16 > /dev/null 2>&1
for t in r03_bf16 r03_bf16x3 r03_train_bf16; do
  cp gpurun_out/prof_$t/summary_$t.txt $O/rocprofv3_${t}_summary.txt
  cp gpurun_out/prof_$t/bench_line_$t.json $O/bench_line_under_rocprof_$t.json
done
cd $R
python bench.py > $O/bench_fwd64_r03.json 2> $O/bench_fwd64.err
python bench.py --workload fwd16_long > $O/bench_fwd16_long_r03.json 2> $O/bench_fwd16_long.err
python bench.py --workload train32 > $O/bench_train32_bf16_r03.json 2> $O/bench_train32.err
python bench.py --workload train32 --precision bf16x3 --no-cpu-baseline > $O/bench_train32_bf16x3_r03.json 2>> $O/bench_train32.err
python bench.py --workload infer64 --no-cpu-baseline > $O/bench_infer64_bf16_r03.json 2> $O/bench_infer.err
python bench.py --workload infer_lj --no-cpu-baseline > $O/bench_infer_lj_bf16_r03.json 2>> $O/bench_infer.err
bash tools/prof_infer.sh > $O/infer_b1.log 2>&1
cp gpurun_out/prof_inf/summary_inf.txt $O/rocprofv3_r03_infer_b1_summary.txt
python tools/gpu_probe_train_host.py > $O/train_host_r03.txt 2>&1
if [ -f lab/rc_stamp.so ]; then
  for sp in 1 2; do PSPLIT=$sp EFTS_LIB=$R/lab/rc_stamp.so timeout 200 python tools/gpu_probe_rc_stamp.py; done > $O/rc_stamps_r03.txt 2>&1
fi
if [ -f lab/rc_phase.so ]; then
  EFTS_LIB=$R/lab/rc_phase.so timeout 200 python tools/gpu_probe_rc_phases.py > $O/rc_phases_r03.txt 2>&1
fi
ls -la $O
