mkdir -p gpurun_out
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5 ) > gpurun_out/t_gpu_all.txt; cat gpurun_out/t_gpu_all.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
