mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -15 ) > gpurun_out/t_train.txt
( timeout 500 bash tools/r06_train_ab.sh ) > gpurun_out/train_ab_r06.txt 2>&1
( timeout 300 bash tools/r06_tile_cost.sh ) > gpurun_out/rc_tile_cost_r06.txt 2>&1
( timeout 300 python -m pytest tests/test_dist_gpu.py -x -q -k "eight" --timeout 140 2>&1 | tail -15 ) > gpurun_out/t_dist8.txt
tail -5 gpurun_out/t_train.txt; cat gpurun_out/train_ab_r06.txt; cat gpurun_out/rc_tile_cost_r06.txt; tail -8 gpurun_out/t_dist8.txt
