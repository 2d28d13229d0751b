# round 5: grouped (stream-K) weight gradients -- tests, then an interleaved A/B of the graphed B=32 step over the workgroup count
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "grouped_wgrad or direct_wgrad or graphed_training or three_training or data_parallel" 2>&1 | tail -5
for rep in 1 2 3; do
for cfg in "_WGRAD_GROUP_WGS=256" "_WGRAD_GROUP_WGS=320" "_WGRAD_GROUP_WGS=384" "_WGRAD_GROUP_WGS=448" "_WGRAD_GROUP_WGS=512"; do
for prec in bf16 bf16x3; do
r=$(timeout 300 python tools/gpu_probe_train_graph.py $prec $cfg 2>&1 | grep "graph" | tail -1 | grep -o "[0-9.]* ms/step")
echo "AB $prec $cfg : $r"
done; done; done 2>&1 | tee gpurun_out/wgrad_ab.txt
