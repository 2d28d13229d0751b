mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "Warn\|warn\|^$" | tail -15 ) > gpurun_out/t_gpu_all.txt; tail -4 gpurun_out/t_gpu_all.txt
S=$(date +%s); timeout 500 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default_r06.json 2> gpurun_out/bench_default_r06.err; echo "bench default rc=$? wall=$(( $(date +%s) - S )) s" > gpurun_out/bench_default_r06.wall; cat gpurun_out/bench_default_r06.wall
( timeout 200 python bench.py --workload vocoder; timeout 200 python bench.py --workload vocoder8 --no-cpu-baseline ) > gpurun_out/bench_vocoder_r06.json 2> gpurun_out/bench_vocoder_r06.err; tail -2 gpurun_out/bench_vocoder_r06.err
S=$(date +%s); timeout 600 python bench.py --workload stock_gpu --stock-find 1 > gpurun_out/bench_stock_gpu_r06.json 2> gpurun_out/bench_stock_gpu_r06.err; echo "stock find wall=$(( $(date +%s) - S )) s"; cut -c1-1200 gpurun_out/bench_stock_gpu_r06.json
