mkdir -p gpurun_out
timeout 600 python tools/micro/layer_yardstick.py > gpurun_out/layer_yardstick.txt 2>gpurun_out/layer_yardstick.err; echo rc=$?
tail -12 gpurun_out/layer_yardstick.txt
cd /tmp && export TMPDIR=/tmp
YROUNDS=1 YLAUNCHES=20 YSUSTAIN=0.05 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/yprof -o y -- python $GRAFT_REPO_ROOT/tools/micro/layer_yardstick.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/yprof -name "*kernel_stats*" | head; f=$(find /tmp/yprof -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/layer_yardstick_kernels.csv; head -30 $f | cut -c1-400
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
