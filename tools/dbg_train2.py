import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import efts_oracle as O
from oracle.precision_emulation import Mode
from efficient_tts_amd import EfficientTTSCNN
from efficient_tts_amd.train import TrainEngine
dev = torch.device("cuda:0")
case = sys.argv[1] if len(sys.argv) > 1 else "fwd_full"
g = np.load(f"tests/golden/{case}.npz")
args = [torch.from_numpy(g[k]) for k in ("text", "text_lengths", "speech", "speech_lengths")]
P = {k: v.clone().requires_grad_(True) for k, v in O.fill_params().items()}
with Mode("bf16x3", "bf16x3"):
    o = O.forward(P, *args, retain=True); o["loss"].backward()
m = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, sigma=0.01); m.load_state_dict(O.fill_params()); m = m.to(dev).eval()
eng = TrainEngine(m)
out3, aux = eng.forward_backward(*[a.to(dev) for a in args], keep=True)
torch.cuda.synchronize()
T2 = args[2].shape[1]
mask = O.non_pad_mask(args[3], T2)
def rel(a, b): return float((a - b).abs().max() / (b.abs().max() + 1e-12))
print("dH", rel(aux["dH"].view().cpu(), o["expanded"].grad.transpose(1, 2) * mask[:, :, None]))
print("dAp", rel(aux["dAp"].cpu() * (o["reconst_alpha"] != 0), o["reconst_alpha"].grad * (o["reconst_alpha"] != 0)))
print("de", rel(aux["de"].cpu(), o["e"].grad), "dpi", rel(aux["dpi"].cpu(), o["imv"].grad))
print("GQ", rel(aux["GQ"].view().cpu(), o["mel_h"].grad), "GK", rel(aux["GK"].view().cpu(), o["text_key"].grad * (o["text_key"] != 0)),
      "GV", rel(aux["GV"].view().cpu(), o["text_value"].grad * (o["text_value"] != 0)))
for n, p in P.items():
    r = p.grad; q = eng.g[n].cpu()
    e = float((q - r).abs().max() / max(float(r.abs().max()), 1e-9))
    if e > 2e-2: print(f"{n:48s} rel {e:.3e}  |ref|max {float(r.abs().max()):.3e} norm got/ref {float(q.norm()):.4e}/{float(r.norm()):.4e}")
