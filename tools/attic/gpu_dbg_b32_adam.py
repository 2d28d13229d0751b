import sys, os, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_fullsize as F
from oracle import efts_oracle as O
from efficient_tts_amd.optim import EftsAdam
from efficient_tts_amd import train as TR
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(3212800)
B, T1, T2 = 32, 128, 800
text = torch.randint(0, 76, (B, T1), generator=gen)
mel = torch.randn(B, T2, 80, generator=gen)
tl = torch.randint(T1 // 2, T1 + 1, (B,), generator=gen); tl[0] = T1
sl = torch.randint(T2 // 2, T2 + 1, (B,), generator=gen); sl[0] = T2
lr = 1e-3
P = {k: v.clone().requires_grad_(True) for k, v in O.fill_params().items()}
params = list(P.values())
opt_ref = torch.optim.Adam(params, lr=lr, betas=(0.9, 0.99), eps=1e-9, weight_decay=1e-5, amsgrad=True)
out = O.forward(P, text, tl, mel, sl)
out["loss"].backward()
g_ref = {k: v.grad.clone() for k, v in P.items()}
gn_ref = float(torch.nn.utils.clip_grad_norm_(params, 1.0))
opt_ref.step()
n = "text_embedding_table.weight"
for fwd, dg in ((0, 0), (3, -1)):
    TR._RESCONV_FWD, TR._RESCONV_DGRAD = fwd, dg
    m = F._model("bf16x3").eval()
    p0 = {k: p.detach().clone() for k, p in m.named_parameters()}
    opt = EftsAdam(m, lr=lr, betas=(0.9, 0.99), eps=1e-9, weight_decay=1e-5, amsgrad=True, grad_norm=1.0)
    loss, stats, *_ = m(text=text.to(dev), text_lengths=tl.to(dev), speech=mel.to(dev), speech_lengths=sl.to(dev))
    opt.zero_grad(); loss.backward()
    g_ours = dict(m.named_parameters())[n].grad.detach().cpu().clone()
    opt.step(); torch.cuda.synchronize()
    a, b = dict(m.named_parameters())[n].detach().cpu(), P[n].detach()
    d = (a - b).abs()
    moved = (p0[n].cpu() - b).abs() > 0.2 * lr
    same = d <= 0.05 * lr
    big = g_ref[n].abs() > 0.1 * g_ref[n].abs().max()
    bad = (big & moved & ~same).nonzero()
    print("cfg", fwd, dg, "bad", bad.tolist()[:5], "max g_ref", float(g_ref[n].abs().max()))
    for i, j in bad.tolist()[:5]:
        print("   ", i, j, "g_ref", float(g_ref[n][i, j]), "g_ours", float(g_ours[i, j]), "p0", float(p0[n][i, j]), "ref", float(b[i, j]), "ours", float(a[i, j]))
