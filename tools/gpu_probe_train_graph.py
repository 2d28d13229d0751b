"""how much of the training step is launch overhead: fwd+bwd captured as one hipGraph (fixed dropout seeds -- a probe, not a product path)"""
import sys, time, torch
sys.path.insert(0, ".")
from efficient_tts_amd import EfficientTTSCNN
from efficient_tts_amd.autograd import engine_of
from efficient_tts_amd.optim import EftsAdam, WarmupLR
from efficient_tts_amd import train as TR
for kv in sys.argv[2:]:                      # module switches of efficient_tts_amd.train: NAME=INT; SKIP=entry,entry: those C-ABI launches become no-ops
    k, v = kv.split("=")                     # (an upper bound of what removing them can return: results are garbage then)
    if k == "SKIP":
        from efficient_tts_amd import lib as _L
        for name in v.split(","):
            assert hasattr(_L.load(), name), name
            setattr(_L.load(), name, lambda *a: 0)
        continue
    assert hasattr(TR, k); setattr(TR, k, int(v))
dev = torch.device("cuda:0")
B, T1, T2 = 32, 128, 800
torch.manual_seed(0)
model = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, use_weighted_masking=False, sigma=0.01, precision=sys.argv[1] if len(sys.argv) > 1 else "bf16").to(dev).train()
opt = EftsAdam(model, lr=1e-3, betas=(0.9, 0.99), eps=1e-9, weight_decay=1e-5, amsgrad=True, grad_norm=1.0)
g = torch.Generator().manual_seed(1234)
text = torch.randint(0, 76, (B, T1), generator=g).to(dev)
mel = torch.randn(B, T2, 80, generator=g).to(dev)
tl = torch.full((B,), T1, dtype=torch.int64, device=dev)
sl = torch.full((B,), T2, dtype=torch.int64, device=dev)
eng = engine_of(model)
def eager():
    loss, *_ = model(text=text, text_lengths=tl, speech=mel, speech_lengths=sl)
    opt.zero_grad(); loss.backward(); opt.step(grad_scale=1.0)
for _ in range(10): eager()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): eager()
torch.cuda.synchronize()
print(f"eager {1e3*(time.perf_counter()-t0)/50:.3f} ms/step")
gr = torch.cuda.CUDAGraph()
model._packed_sig = None
with torch.cuda.graph(gr, capture_error_mode="thread_local"):
    out3, _ = eng.forward_backward(text, tl, mel, sl)
    model.dropout_calls -= 1
    opt.step(grad_scale=1.0)
for _ in range(5): gr.replay()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(50): gr.replay()
    torch.cuda.synchronize()
    print(f"graph (fwd+bwd+adam) {1e3*(time.perf_counter()-t0)/50:.3f} ms/step  loss {float(out3[0]):.4f}")
