cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_inf
mkdir -p $OUT
python $R/tools/gpu_trace_infer.py 200
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python $R/tools/gpu_trace_infer.py 5 > $OUT/trace.log 2>&1
TIMELINE=60 python $R/tools/prof_summary.py $OUT inf > $OUT/summary_inf.txt 2>&1
find $OUT -name "*.db" -delete
