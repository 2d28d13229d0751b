# round 5: dgrad launch with the fused activation backward -- tests, then an interleaved A/B of the graphed B=32 step
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_resconv_gpu.py -x -q -m gpu -k "fused_activation or dgrad_launch or sign_bits or graphed_training or full_size_param or resconv" 2>&1 | tail -8
for rep in 1 2 3; do
for cfg in "_FUSE_ACT_BWD=0" "_FUSE_ACT_BWD=1" "_FUSE_ACT_BWD=0 _RESCONV_DGRAD=3" "_FUSE_ACT_BWD=1 _RESCONV_DGRAD=3"; do
for prec in bf16 bf16x3; do
r=$(timeout 300 python tools/gpu_probe_train_graph.py $prec $cfg 2>&1 | grep "graph" | tail -1 | grep -o "[0-9.]* ms/step")
echo "AB $prec $cfg : $r"
done; done; done 2>&1 | tee gpurun_out/fuse_ab.txt
