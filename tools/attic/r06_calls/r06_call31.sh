mkdir -p gpurun_out
B="python bench.py --workload train32 --precision bf16 --no-cpu-baseline --measure-traffic 0"
P='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print("graph", round(d["graph_ms_per_step"],4), "eager", round(d["eager_ms_per_step"],4))'
for rep in 1 2; do
echo -n "default: "; timeout 300 $B 2>/dev/null | python -c "$P"
for q in 1 2 3 4 6 8; do echo -n "DEBUG_HIP_FORCE_GRAPH_QUEUES=$q: "; DEBUG_HIP_FORCE_GRAPH_QUEUES=$q timeout 300 $B 2>/dev/null | python -c "$P"; done
echo -n "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0: "; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 300 $B 2>/dev/null | python -c "$P"
done > gpurun_out/graph_queues.txt 2>&1; cat gpurun_out/graph_queues.txt
