mkdir -p gpurun_out
for i in 1 2; do
( timeout 300 python -m pytest tests/test_gpu_train.py -x -q --tb=long 2>&1 | grep -v "Warn\|warn\|^$" | tail -70 ) > gpurun_out/t_train_full_$i.txt
tail -3 gpurun_out/t_train_full_$i.txt
done
