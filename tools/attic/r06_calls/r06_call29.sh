mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -4 ) > gpurun_out/t_train.txt; cat gpurun_out/t_train.txt
for rep in 1 2; do for v in 1 0; do timeout 300 python bench.py --workload train32 --precision bf16 --no-cpu-baseline --measure-traffic 0 --train-set _GV_ON_SIDE=$v 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('_GV_ON_SIDE=$v bf16 graph', round(d['graph_ms_per_step'],4), 'eager', round(d['eager_ms_per_step'],4))"; done; done > gpurun_out/train_gv_side.txt 2>&1
cat gpurun_out/train_gv_side.txt
