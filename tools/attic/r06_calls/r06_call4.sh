mkdir -p gpurun_out
( timeout 500 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | grep -v "Warn\|warn" | tail -12 ) > gpurun_out/t_train.txt
( timeout 500 bash tools/r06_train_ab.sh ) > gpurun_out/train_ab2_r06.txt 2>&1
( timeout 300 python -m pytest tests/test_dist_gpu.py -x -q -k "eight" --timeout 200 --tb=short 2>&1 | grep -v "Warn\|warn" | tail -40 ) > gpurun_out/t_dist8.txt
S=$(date +%s); timeout 500 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default_r06.json 2> gpurun_out/bench_default_r06.err; echo "bench default rc=$? wall=$(( $(date +%s) - S )) s" > gpurun_out/bench_default_r06.wall
( timeout 200 python bench.py --workload vocoder --no-cpu-baseline; timeout 200 python bench.py --workload vocoder8 --no-cpu-baseline ) > gpurun_out/bench_vocoder_r06.json 2> gpurun_out/bench_vocoder_r06.err
tail -6 gpurun_out/t_train.txt; cat gpurun_out/train_ab2_r06.txt; tail -30 gpurun_out/t_dist8.txt; cat gpurun_out/bench_default_r06.wall; tail -3 gpurun_out/bench_default_r06.err; cut -c1-1500 gpurun_out/bench_vocoder_r06.json; tail -3 gpurun_out/bench_vocoder_r06.err
