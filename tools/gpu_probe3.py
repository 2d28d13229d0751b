import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficient_tts_amd import EfficientTTSCNN
from efficient_tts_amd.optim import EftsAdam, WarmupLR
dev = torch.device("cuda:0")
B, T1, T2 = 32, 128, 800
m = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, sigma=0.01, precision=sys.argv[1] if len(sys.argv) > 1 else "bf16").to(dev).train()
opt = EftsAdam(m); sch = WarmupLR(opt, 4000)
g = torch.Generator().manual_seed(1)
text = torch.randint(0, 76, (B, T1), generator=g).to(dev); mel = torch.randn(B, T2, 80, generator=g).to(dev)
tl = torch.full((B,), T1, dtype=torch.int64, device=dev); sl = torch.full((B,), T2, dtype=torch.int64, device=dev)
def step():
    loss, stats, *_ = m(text=text, text_lengths=tl, speech=mel, speech_lengths=sl)
    opt.zero_grad(); loss.backward(); opt.step(); sch.step()
for _ in range(3): step()
torch.cuda.synchronize()
# host-only cost: time the enqueue without waiting
t0 = time.perf_counter()
for _ in range(10): step()
t_host = (time.perf_counter() - t0) / 10
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 10
print(f"enqueue {t_host*1e3:.2f} ms/step, wall {t_all*1e3:.2f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
