import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from efficient_tts_amd.frontend import LogMelFrontend
from oracle import logmel_oracle as O
dev = torch.device("cuda:0")
def synth_audio(B, L, seed):
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(L, dtype=torch.float64) / 22050
    out = []
    for b in range(B):
        f0 = 90 + 140 * torch.rand(1, generator=g).item()
        y = torch.zeros(L, dtype=torch.float64)
        for h in range(1, 30):
            y += torch.sin(2 * np.pi * f0 * h * t + 6.28 * torch.rand(1, generator=g).item()) / h
        env = (0.5 + 0.5 * torch.sin(2 * np.pi * 3.1 * t + b)).clamp(min=0) ** 2
        y = 0.25 * y * env + 0.01 * torch.randn(L, generator=g, dtype=torch.float64)
        y[L // 3: L // 3 + 3000] = 0.0003 * torch.randn(3000, generator=g, dtype=torch.float64)   # near-silence
        y[L // 2: L // 2 + 2000] = 0.0                                                             # digital silence
        out.append(y)
    a = torch.stack(out).clamp(-1, 1)
    return torch.round(a * 32767).to(torch.int16)
B, Lm = 4, 40000
a16 = synth_audio(B, Lm, 0)
lengths = torch.tensor([40000, 33333, 25601, 12800])
fe = LogMelFrontend(dev)
mel, frames = fe(a16, lengths)
ref, rfr = O.batch_logmel(a16.float() / 32768.0, lengths)
print("frames", frames.tolist(), rfr.tolist(), mel.shape, ref.shape)
d = (mel.cpu() - ref).abs()
print("max abs err", d.max().item(), "mean", d.mean().item())
for thr in (-11.0, -9, -7, -5, -3):
    m = ref > thr
    print(f" where ref > {thr}: max err {d[m].max().item():.2e}  frac {m.float().mean().item():.3f}")
idx = torch.nonzero(d == d.max())[0]; print("worst at", idx.tolist(), mel.cpu()[tuple(idx)].item(), ref[tuple(idx)].item())
# timing at B=64 x 800 frames
a = torch.randn(64, 800 * 256, device=dev).clamp(-1, 1) * 0.3
ln = torch.full((64,), 800 * 256)
for _ in range(3): fe(a, ln)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): fe(a, ln)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print(f"B=64 x 800 frames: {dt*1e3:.3f} ms  {64*800/dt/1e6:.1f} M frames/s")
