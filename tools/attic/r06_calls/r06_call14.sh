mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mel_loss or narrow_tiles or forward or gemm_conv" 2>&1 | grep -v "Warn\|warn\|^$" | tail -8 ) > gpurun_out/t_parity.txt; tail -3 gpurun_out/t_parity.txt
B="python bench.py --no-cpu-baseline --parity-mode 0 --call-modes 0 --measure-traffic 0 --train-record 0 --sub-records 0 --stock-gpu 0"
P='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["ms_per_step"],4), "ms", d.get("hip_vs_oracle_mel_max_abs"))'
for rep in 1 2; do
for c in 128 144 160 176; do echo -n "bf16 prenet_cus=$c "; EFTS_AB_PRENET_CUS=$c timeout 300 $B --precision bf16 2>/dev/null | python -c "$P"; done
for c in 128 160; do echo -n "bf16x3 prenet_cus=$c "; EFTS_AB_PRENET_CUS=$c timeout 300 $B --precision bf16x3 2>/dev/null | python -c "$P"; done
done > gpurun_out/fwd_ab.txt 2>&1; cat gpurun_out/fwd_ab.txt
