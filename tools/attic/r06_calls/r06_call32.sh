mkdir -p gpurun_out
B="python bench.py --workload train32 --precision bf16 --no-cpu-baseline --measure-traffic 0"
P='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print("graph", round(d["graph_ms_per_step"],4), "eager", round(d["eager_ms_per_step"],4))'
for rep in 1 2; do
for q in 4 8 16; do for g in 1 2; do echo -n "QUEUES=$q _GV_ON_SIDE=$g: "; DEBUG_HIP_FORCE_GRAPH_QUEUES=$q timeout 300 $B --train-set _GV_ON_SIDE=$g 2>/dev/null | python -c "$P"; done; done
done > gpurun_out/graph_queues2.txt 2>&1; cat gpurun_out/graph_queues2.txt
