#!/bin/bash
# Round-6 evidence in one gpurun call: lab libraries rebuilt from HEAD, workgroup stamps / phase counters / tile marks / ablations of the
# dominant kernel, rocprofv3 summaries (trace + PMC passes, with timelines for BOTH precisions), the bench lines of every workload.
# Everything lands in gpurun_out/collect/; copy what is to be judged into profiles/.  A failing probe fails the script.
set -eo pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/collect
mkdir -p $O
cd $R
TAGR=${TAGR:-r06}
if [ -z "$SKIP_LAB" ]; then
EXPS="1 8 9 4" bash tools/lab_build.sh > $O/lab_build.log 2>&1
probe() {   # name, lab library, script, env...
  local out=$O/$1; local lib=$2; local script=$3; shift 3
  env "$@" EFTS_LIB=$R/lab/$lib timeout 300 python $script > $out 2>&1 || { echo "PROBE FAILED: $out"; tail -5 $out; exit 1; }
  if grep -q "Traceback" $out; then echo "PROBE FAILED (traceback): $out"; exit 1; fi
}
( for sp in 1 2; do PSPLIT=$sp EFTS_LIB=$R/lab/rc_stamp.so timeout 300 python tools/gpu_probe_rc_stamp.py; done ) > $O/rc_stamps_$TAGR.txt 2>&1 || { echo "PROBE FAILED: stamps"; exit 1; }
if grep -q Traceback $O/rc_stamps_$TAGR.txt; then echo "PROBE FAILED (traceback): stamps"; exit 1; fi
probe rc_phases_$TAGR.txt rc_phase.so tools/gpu_probe_rc_phases.py
probe rc_marks_$TAGR.txt rc_marks.so tools/gpu_probe_rc_marks.py
# the same marks for the one-wave-per-SIMD kernel (bf16 planes only: RCK=2 falls back to the 8-wave kernel for split 2), and its micro benchmark
probe rc_marks_w4_$TAGR.txt rc_marks.so tools/gpu_probe_rc_marks.py RCK=2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I efficient_tts_amd/csrc tools/micro/rc4_loop_test.hip -o lab/rc4_loop_test > $O/rc4_micro_build.log 2>&1 && timeout 200 lab/rc4_loop_test > $O/rc4_loop_micro_$TAGR.txt 2>&1 || { echo "PROBE FAILED: rc4 micro benchmark"; exit 1; }
# what the part gives a bare MFMA stream: zero / random operands, all CUs / one CU (TFLOP/s and the delivered shader clock)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_ceiling.hip -o lab/mfma_ceiling > /dev/null 2>&1 && timeout 200 lab/mfma_ceiling > $O/mfma_ceiling_$TAGR.txt 2>&1 || { echo "PROBE FAILED: mfma ceiling"; exit 1; }
# ablations of the ping-pong kernel (us per B = 64 launch): all / no LDS-DMA / no epilogue traffic / neither / no MFMA + fragment reads
GRAFT_REPO_ROOT=$R bash tools/rc_ab.sh 64x800 cur exp1 exp8 exp9 exp4 > $O/rc_ablate_$TAGR.txt 2>&1 || { echo "PROBE FAILED: ablations"; exit 1; }
# energy per launch (rocm-smi socket power beside a sustained loop of the layer x its time per launch): both kernels, both precisions
bash tools/gpu_power_trace.sh > $O/power_trace_$TAGR.txt 2>&1 || { echo "PROBE FAILED: power trace"; exit 1; }
python tools/power_summary.py $O/power_trace_$TAGR.txt >> $O/rc_ablate_$TAGR.txt
fi
if [ -z "$SKIP_PROF" ]; then
PREC=bf16 TIMELINE=60 bash tools/prof_conv.sh ${TAGR}_bf16 > /dev/null 2>&1
PREC=bf16x3 TIMELINE=60 bash tools/prof_conv.sh ${TAGR}_bf16x3 > /dev/null 2>&1
WL=train32 STEPS=3 TSTEPS=3 TWARM=2 TIMELINE=240 BARGS="--train-graph 0" bash tools/prof_conv.sh ${TAGR}_train_bf16 > /dev/null 2>&1
# the same training step as ONE hipGraph replay (what bench.py times): trace only, nothing issued behind the replays
# (round 6: WITH the counter passes -- rocprofv3 serialises a replayed graph's kernels for them, which is what the per-kernel counters want)
EFTS_BENCH_TRAIN_NO_EAGER=1 WL=train32 STEPS=4 TSTEPS=20 TWARM=5 TIMELINE=215 BARGS="--train-graph 1" bash tools/prof_conv.sh ${TAGR}_train_graph > /dev/null 2>&1
# the frozen "next" rows: HiFi-GAN generator (one utterance, a batch of 8) and the log-mel front-end, trace only
NOPMC=1 WL=vocoder STEPS=5 TSTEPS=20 TWARM=5 TIMELINE=100 bash tools/prof_conv.sh ${TAGR}_vocoder > /dev/null 2>&1
NOPMC=1 WL=vocoder8 STEPS=5 TSTEPS=10 TWARM=3 TIMELINE=100 bash tools/prof_conv.sh ${TAGR}_vocoder8 > /dev/null 2>&1
NOPMC=1 WL=logmel64 STEPS=5 TSTEPS=20 TWARM=5 TIMELINE=12 bash tools/prof_conv.sh ${TAGR}_logmel > /dev/null 2>&1
for t in ${TAGR}_bf16 ${TAGR}_bf16x3 ${TAGR}_train_bf16 ${TAGR}_train_graph ${TAGR}_vocoder ${TAGR}_vocoder8 ${TAGR}_logmel; do
  test -s gpurun_out/prof_$t/summary_$t.txt || { echo "PROFILE FAILED: $t"; exit 1; }
  cp gpurun_out/prof_$t/summary_$t.txt $O/rocprofv3_${t}_summary.txt
  cp gpurun_out/prof_$t/bench_line_$t.json $O/bench_line_under_rocprof_$t.json
done
fi
cd $R
if [ -z "$SKIP_BENCH" ]; then
python bench.py > $O/bench_fwd64_$TAGR.json 2> $O/bench_fwd64.err
python bench.py --workload fwd16_long > $O/bench_fwd16_long_$TAGR.json 2> $O/bench_fwd16_long.err
python bench.py --workload train32 > $O/bench_train32_bf16_$TAGR.json 2> $O/bench_train32.err
python bench.py --workload train32 --precision bf16x3 --no-cpu-baseline > $O/bench_train32_bf16x3_$TAGR.json 2>> $O/bench_train32.err
python bench.py --workload infer64 > $O/bench_infer64_bf16_$TAGR.json 2> $O/bench_infer.err
python bench.py --workload infer_lj > $O/bench_infer_lj_bf16_$TAGR.json 2>> $O/bench_infer.err
python bench.py --workload logmel64 > $O/bench_logmel64_$TAGR.json 2>> $O/bench_infer.err
python bench.py --workload vocoder --no-cpu-baseline > $O/bench_vocoder_$TAGR.json 2>> $O/bench_infer.err
# launch-skipping bounds of the forward (what removing a group of launches could return at most), both precisions
( timeout 300 python tools/gpu_probe_fwd_skip.py bf16; timeout 300 python tools/gpu_probe_fwd_skip.py bf16x3 ) 2>&1 | grep BOUND > $O/fwd_skip_bounds_$TAGR.txt
# the forward on the one-wave-per-SIMD kernel (where it is eligible), interleaved with the default for an in-situ comparison
for k in 0 2 0 2; do python bench.py --rc-kernel $k --no-cpu-baseline --parity-mode 0 --call-modes 0 --measure-traffic 0 --train-record 0 --sub-records 0 --stock-gpu 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('rc-kernel $k', 'ms_per_step', round(d['ms_per_step'], 4), 'roofline', d['roofline']['achieved'], d['roofline']['frac'])"; done > $O/bench_rc_kernel_ab_$TAGR.txt 2>&1
for f in $O/bench_*_$TAGR.json; do test -s $f || { echo "BENCH FAILED: $f"; exit 1; }; done
fi
ls -la $O
