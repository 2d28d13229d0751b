mkdir -p gpurun_out
timeout 600 python tools/gpu_ab_forward.py base= nofuse=fuse_mel_loss:0 2>&1 | grep forward > gpurun_out/ab_mel_loss.txt; cat gpurun_out/ab_mel_loss.txt
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5 ) > gpurun_out/t_gpu_all.txt; cat gpurun_out/t_gpu_all.txt
