"""Soak: a few hundred optimizer steps of the fused training pass on one fixed synthetic batch (over-fit test): the loss must
fall monotonically-ish and stay finite, in both precisions, with and without conv dropout.  Catches rare wrong tiles / races
that single-step parity tests cannot."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficient_tts_amd import EfficientTTSCNN
from efficient_tts_amd.optim import EftsAdam, WarmupLR
dev = torch.device("cuda:0")
B, T1, T2 = int(os.environ.get("PB", 32)), 128, 800
g = torch.Generator().manual_seed(7)
text = torch.randint(1, 76, (B, T1), generator=g).to(dev)
mel = (-4.0 + 2.0 * torch.randn(B, T2, 80, generator=g)).clamp(-11.5, 2.0).to(dev)
tl = torch.randint(T1 // 2, T1 + 1, (B,), generator=g).to(dev); tl[0] = T1
sl = torch.randint(T2 // 2, T2 + 1, (B,), generator=g).to(dev); sl[0] = T2
for b in range(B):
    text[b, tl[b]:] = 0; mel[b, sl[b]:] = 0
from efficient_tts_amd.step_graph import GraphedStep
for prec, drop, graphed in (("bf16", 0.0, False), ("bf16", 0.0, True), ("bf16x3", 0.0, True), ("bf16", 0.1, False)):
    torch.manual_seed(0)
    m = EfficientTTSCNN(num_symbols=76, dropout_rate=drop, use_masking=True, sigma=0.01, precision=prec).to(dev).train()
    opt = EftsAdam(m, lr=1e-3, betas=(0.9, 0.99), eps=1e-9, weight_decay=1e-5, amsgrad=True, grad_norm=1.0)
    sch = WarmupLR(opt, warmup_steps=50)
    hist = []
    gs = GraphedStep(m, opt, sch)
    for step in range(int(os.environ.get("STEPS", 300))):
        if graphed:                                        # one hipGraph replay per step (the mel-length stacks run on efts_resconv5 at this size)
            loss, stats = gs(text, tl, mel, sl)
        else:
            loss, stats, *_ = m(text=text, text_lengths=tl, speech=mel, speech_lengths=sl)
            opt.zero_grad(); loss.backward(); opt.step(); sch.step()
        if step % 50 == 0 or step == int(os.environ.get("STEPS", 300)) - 1:
            hist.append((step, float(loss)))
    ok = all(v == v and v < 1e4 for _, v in hist) and hist[-1][1] < 0.5 * hist[0][1]
    assert (gs.replays == int(os.environ.get("STEPS", 300)) - 1) == graphed
    print(f"{prec} dropout {drop} {'graphed' if graphed else 'eager'}: " + "  ".join(f"{s}:{v:.4f}" for s, v in hist) + ("  OK" if ok else "  FAILED"), flush=True)
