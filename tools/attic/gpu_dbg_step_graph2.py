import sys, torch
sys.path.insert(0, ".")
from efficient_tts_amd import EfficientTTSCNN
from efficient_tts_amd.optim import EftsAdam, WarmupLR
from efficient_tts_amd.step_graph import GraphedStep
dev = torch.device("cuda:0")
B, T1, T2 = (int(x) for x in sys.argv[1].split(","))
gen = torch.Generator().manual_seed(B * 31 + T2)
batches = []
for _ in range(2):
    batches.append((torch.randint(0, 76, (B, T1), generator=gen).to(dev), torch.randint(T1 // 2, T1 + 1, (B,), generator=gen).to(dev),
                    torch.randn(B, T2, 80, generator=gen).to(dev), torch.randint(T2 // 2, T2 + 1, (B,), generator=gen).to(dev)))
print([b[1].tolist() for b in batches], [b[3].tolist() for b in batches])
torch.manual_seed(1)
m = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, sigma=0.01, precision="bf16").to(dev).train()
opt = EftsAdam(m, lr=1e-3, betas=(0.9, 0.99), eps=1e-9, weight_decay=1e-5, amsgrad=True, grad_norm=1.0)
sch = WarmupLR(opt, warmup_steps=50)
step = GraphedStep(m, opt, sch)
for i in range(6):
    a = batches[i % 2]
    print("step", i, flush=True)
    loss, stats = step._eager(*a)
    torch.cuda.synchronize()
    print("   loss", float(loss), flush=True)
    if i == 3:
        m.eval()
        with torch.no_grad():
            print("   eval", float(m(*a)[0]), flush=True)
        m.train()
