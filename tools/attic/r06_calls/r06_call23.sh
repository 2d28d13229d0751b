mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_frontend.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert|radix" | tail -12 ) > gpurun_out/t_fe.txt; cat gpurun_out/t_fe.txt
