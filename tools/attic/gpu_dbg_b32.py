"""B = 32 training step, gradients against the oracle's autograd, with the mel-length stacks on efts_gemm / on efts_resconv5"""
import sys, os, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from efficient_tts_amd.train import TrainEngine
from efficient_tts_amd import train as TR
import test_gpu_fullsize as F
from oracle import efts_oracle as O
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(3212800)
B, T1, T2 = 32, 128, 800
text = torch.randint(0, 76, (B, T1), generator=gen)
mel = torch.randn(B, T2, 80, generator=gen)
tl = torch.randint(T1 // 2, T1 + 1, (B,), generator=gen); tl[0] = T1
sl = torch.randint(T2 // 2, T2 + 1, (B,), generator=gen); sl[0] = T2
P = {k: v.clone().requires_grad_(True) for k, v in O.fill_params().items()}
out = O.forward(P, text, tl, mel, sl)
out["loss"].backward()
g_ref = {k: v.grad.clone() for k, v in P.items()}
for prec in ("bf16x3",):
    m = F._model(prec).eval()
    eng = TrainEngine(m)
    for name, fwd, dg in (("gemm", 0, 0), ("resconv fwd", 3, 0), ("resconv fwd+dgrad", 3, 3)):
        TR._RESCONV_FWD, TR._RESCONV_DGRAD = fwd, dg
        o3, _ = eng.forward_backward(text.to(dev), tl.to(dev), mel.to(dev), sl.to(dev)); torch.cuda.synchronize()
        rows = []
        for n in g_ref:
            if n == "text_encoder_key.bias": continue
            got = eng.g[n].cpu()
            rows.append((float((got - g_ref[n]).abs().max()) / float(g_ref[n].abs().max()), float((got - g_ref[n]).double().norm() / g_ref[n].double().norm()), n))
        rows.sort(reverse=True)
        print(prec, name, "loss", float(o3[0]), "oracle", float(out["loss"]))
        for r in rows[:5]: print(f"    el {r[0]:.2e} norm {r[1]:.2e} {r[2]}")
        e = [r for r in rows if r[2] == "text_embedding_table.weight"][0]
        print(f"    embedding: el {e[0]:.2e} norm {e[1]:.2e}")
