import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import efts_oracle as O
from efficient_tts_amd import EfficientTTSCNN
from efficient_tts_amd.train import TrainEngine
dev = torch.device("cuda:0")
g = np.load("tests/golden/fwd_tiny.npz")
args = [torch.from_numpy(g[k]) for k in ("text", "text_lengths", "speech", "speech_lengths")]
P = {k: v.clone().requires_grad_(True) for k, v in O.fill_params().items()}
o = O.forward(P, *args, retain=True); o["loss"].backward()
m = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, sigma=0.01); m.load_state_dict(O.fill_params()); m = m.to(dev).eval()
eng = TrainEngine(m)
out3, aux = eng.forward_backward(*[a.to(dev) for a in args], keep=True)
torch.cuda.synchronize()
T2 = args[2].shape[1]
mask = O.non_pad_mask(args[3], T2)
ref = o["expanded"].grad.transpose(1, 2) * mask[:, :, None]
got = aux["dH"].view().cpu()
d = (got - ref).abs()
print("max ref", ref.abs().max().item(), "max err", d.max().item())
idx = torch.nonzero(d > 0.2 * d.max())
print("n big", len(idx), idx[:20].tolist())
print("err by j (item0):", d[0].max(dim=1).values[:64].tolist())
print("err by j (item1):", d[1].max(dim=1).values[:64].tolist())
for n in ["decoder.layers.5.conv.0.bias", "decoder.layers.0.conv.0.bias", "mel_output_layer.weight", "decoder.layers.5.conv.0.weight_v", "decoder.layers.5.conv.0.weight_g"]:
    r = P[n].grad; q = eng.g[n].cpu()
    print(n, "rel", float((q - r).abs().max() / r.abs().max()))
