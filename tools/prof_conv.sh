#!/bin/bash
# rocprofv3 recipes for the hot path (run on the GPU box via gpurun).  Counters are collected in
# their own passes (--kernel-trace only), each pass under its own timeout; summaries are written
# to gpurun_out/prof_<tag>/ by tools/prof_summary.py and then copied into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r01}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
BENCH="python $R/bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline --parity-mode 0 --call-modes 0 --measure-traffic 0 --train-record 0 --sub-records 0 --stock-gpu 0 --precision ${PREC:-bf16} --workload ${WL:-fwd64} ${BARGS:-}"
# the trace pass runs long enough for the clock governor to settle (30 + 10 steps: short samples run 3-5 % slower, DESIGN.md 4a'); the
# counter passes below keep the short run
TBENCH=${BENCH/--steps ${STEPS:-5} --warmup 2/--steps ${TSTEPS:-30} --warmup ${TWARM:-10}}
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $TBENCH > $OUT/trace.log 2>&1
if [ -z "$NOPMC" ]; then
timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/pmc_sq -o bench -- $BENCH > $OUT/pmc_sq.log 2>&1
timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT/pmc_sq2 -o bench -- $BENCH > $OUT/pmc_sq2.log 2>&1
timeout -k 5 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o bench -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout -k 5 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc_write -o bench -- $BENCH > $OUT/pmc_write.log 2>&1
timeout -k 5 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_l2 -o bench -- $BENCH > $OUT/pmc_l2.log 2>&1
fi
grep -h '^{"metric' $OUT/trace.log > $OUT/bench_line_${TAG}.json
python $R/tools/prof_summary.py $OUT $TAG > $OUT/summary_${TAG}.txt 2>&1
cat $OUT/summary_${TAG}.txt
find $OUT -name "*.db" -delete
