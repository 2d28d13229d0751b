"""Training step B=32: host enqueue time per step (no synchronisation inside) against the device time per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficient_tts_amd import EfficientTTSCNN
from efficient_tts_amd.optim import EftsAdam, WarmupLR
dev = torch.device("cuda:0")
B, T1, T2 = 32, 128, 800
torch.manual_seed(0)
model = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, use_weighted_masking=False, sigma=0.01, precision="bf16").to(dev).train()
opt = EftsAdam(model, lr=1e-3, betas=(0.9, 0.99), eps=1e-9, weight_decay=1e-5, amsgrad=True, grad_norm=1.0)
sch = WarmupLR(opt, warmup_steps=4000)
g = torch.Generator().manual_seed(1234)
text = torch.randint(0, 76, (B, T1), generator=g).to(dev); mel = torch.randn(B, T2, 80, generator=g).to(dev)
tl = torch.full((B,), T1, dtype=torch.int64, device=dev); sl = torch.full((B,), T2, dtype=torch.int64, device=dev)
def step():
    loss, stats, *_ = model(text=text, text_lengths=tl, speech=mel, speech_lengths=sl)
    opt.zero_grad(); loss.backward(); opt.step(); sch.step()
    return loss
for _ in range(5): step()
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(20):
    h0 = time.perf_counter(); step(); host.append(time.perf_counter() - h0)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print(f"device-paced step {dt * 1e3:.3f} ms; host enqueue per step: median {sorted(host)[10] * 1e3:.3f} ms, min {min(host) * 1e3:.3f}, max {max(host) * 1e3:.3f}")
# the same with a sync before each step: pure host time when the queue is empty
host2 = []
for _ in range(10):
    torch.cuda.synchronize(); h0 = time.perf_counter(); step(); host2.append(time.perf_counter() - h0)
torch.cuda.synchronize()
print(f"host enqueue per step with an empty queue: median {sorted(host2)[5] * 1e3:.3f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
