#!/bin/bash
# kernel-trace only (fast): usage  WL=train32 PREC=bf16 bash tools/prof_trace.sh tag
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-trace}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --precision ${PREC:-bf16} --workload ${WL:-fwd64} > $OUT/trace.log 2>&1
grep -h '^{"metric' $OUT/trace.log > $OUT/bench_line_${TAG}.json
python $R/tools/prof_summary.py $OUT $TAG > $OUT/summary_${TAG}.txt 2>&1
head -40 $OUT/summary_${TAG}.txt | cut -c1-150
find $OUT -name "*.db" -delete
