"""Training step B=32: device time of the segments of the fused forward + backward on the main stream (events at the engine's
marks), un-profiled, median of 20 steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from efficient_tts_amd import EfficientTTSCNN
from efficient_tts_amd.autograd import engine_of
from efficient_tts_amd.optim import EftsAdam, WarmupLR
dev = torch.device("cuda:0")
B, T1, T2 = 32, 128, 800
torch.manual_seed(0)
model = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, use_weighted_masking=False, sigma=0.01, precision=os.environ.get("PREC", "bf16")).to(dev).train()
opt = EftsAdam(model, lr=1e-3, betas=(0.9, 0.99), eps=1e-9, weight_decay=1e-5, amsgrad=True, grad_norm=1.0)
g = torch.Generator().manual_seed(1234)
text = torch.randint(0, 76, (B, T1), generator=g).to(dev); mel = torch.randn(B, T2, 80, generator=g).to(dev)
tl = torch.full((B,), T1, dtype=torch.int64, device=dev); sl = torch.full((B,), T2, dtype=torch.int64, device=dev)
eng = engine_of(model)
marks = []
def mark(name):
    ev = torch.cuda.Event(enable_timing=True); ev.record(); marks.append((name, ev))
def step():
    loss, *_ = model(text=text, text_lengths=tl, speech=mel, speech_lengths=sl)
    opt.zero_grad(); loss.backward(); mark("backward_returned"); opt.step(); mark("optimizer_done")
for _ in range(5): step()
eng.mark = mark
rows = []
for _ in range(20):
    marks.clear()
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    step()
    torch.cuda.synchronize()
    prev, seg = e0, {}
    for name, ev in marks:
        seg[name] = prev.elapsed_time(ev) * 1e3; prev = ev
    rows.append(seg)
names = [n for n, _ in marks]
print("segment (ends at mark): median us over 20 steps")
tot = 0.0
for n in names:
    v = sorted(r[n] for r in rows)[10]; tot += v
    print(f"  {n:26s} {v:8.1f}")
print(f"  {'sum':26s} {tot:8.1f}")
