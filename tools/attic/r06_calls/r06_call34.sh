mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_train.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -4 ) > gpurun_out/t_train.txt; cat gpurun_out/t_train.txt
B="python bench.py --workload train32 --no-cpu-baseline --measure-traffic 0"
P='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print("graph", round(d["graph_ms_per_step"],4), "eager", round(d["eager_ms_per_step"],4))'
for rep in 1 2 3; do for v in 1 0; do echo -n "_ALIGN_BWD_FUSED=$v bf16: "; timeout 300 $B --precision bf16 --train-set _ALIGN_BWD_FUSED=$v 2>/dev/null | python -c "$P"; done; done > gpurun_out/train_align_fused.txt 2>&1
for v in 1 0 1 0; do echo -n "_ALIGN_BWD_FUSED=$v bf16x3: "; timeout 300 $B --precision bf16x3 --train-set _ALIGN_BWD_FUSED=$v 2>/dev/null | python -c "$P"; done >> gpurun_out/train_align_fused.txt 2>&1
cat gpurun_out/train_align_fused.txt
