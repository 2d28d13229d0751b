import sys, torch
sys.path.insert(0, ".")
from efficient_tts_amd import EfficientTTSCNN
from efficient_tts_amd.optim import EftsAdam, WarmupLR
from efficient_tts_amd.step_graph import GraphedStep
dev = torch.device("cuda:0")
B, T1, T2 = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "3,40,130").split(","))
gen = torch.Generator().manual_seed(7)
a = (torch.randint(0, 76, (B, T1), generator=gen).to(dev), torch.randint(T1 // 2, T1 + 1, (B,), generator=gen).to(dev),
     torch.randn(B, T2, 80, generator=gen).to(dev), torch.randint(T2 // 2, T2 + 1, (B,), generator=gen).to(dev))
mode = sys.argv[2] if len(sys.argv) > 2 else ""
for graphed in (False, False, True, True):
    torch.manual_seed(1)
    m = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, sigma=0.01, precision="bf16").to(dev).train()
    opt = EftsAdam(m, lr=1e-3, betas=(0.9, 0.99), eps=1e-9, weight_decay=1e-5, amsgrad=True, grad_norm=1.0)
    sch = None if "nosch" in mode else WarmupLR(opt, warmup_steps=10)
    if "eval" in mode: m.eval()
    if "noclip" in mode: opt.grad_norm = 0.0
    step = GraphedStep(m, opt, sch)
    for i in range(5):
        loss, _ = step(*a) if graphed else step._eager(*a)
        torch.cuda.synchronize()
        print("graphed" if graphed else "eager", i, float(loss), "replays", step.replays, flush=True)
