mkdir -p gpurun_out
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
for rep in 1 2; do for v in none 1 -1; do if [ $v = none ]; then unset EFTS_AB_SIDE_PRIO; else export EFTS_AB_SIDE_PRIO=$v; fi; timeout 300 python tools/gpu_ab_forward.py base= 2>&1 | grep forward | sed "s/base/side prio $v/"; done; done > gpurun_out/ab_side_prio.txt; cat gpurun_out/ab_side_prio.txt
