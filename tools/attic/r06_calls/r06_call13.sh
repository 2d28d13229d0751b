mkdir -p gpurun_out
for rep in 1 2; do
echo "== old ring (A one step ahead)"; EFTS_LIB=$GRAFT_REPO_ROOT/lab/libefts_old.so timeout 300 python tools/gpu_probe_gemm1.py 2>&1 | grep -v "Warn\|amdgpu.ids"
echo "== new ring (A two steps ahead)"; timeout 300 python tools/gpu_probe_gemm1.py 2>&1 | grep -v "Warn\|amdgpu.ids"
done > gpurun_out/gemm1_probe.txt; cat gpurun_out/gemm1_probe.txt
