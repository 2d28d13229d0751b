import sys, torch
sys.path.insert(0, ".")
from efficient_tts_amd import EfficientTTSCNN, ops as O
from efficient_tts_amd.autograd import engine_of
dev = torch.device("cuda:0")
B, T1, T2 = 3, 40, 130
gen = torch.Generator().manual_seed(7)
a = (torch.randint(0, 76, (B, T1), generator=gen).to(dev), torch.randint(T1 // 2, T1 + 1, (B,), generator=gen).to(dev),
     torch.randn(B, T2, 80, generator=gen).to(dev), torch.randint(T2 // 2, T2 + 1, (B,), generator=gen).to(dev))
torch.manual_seed(1)
m = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, sigma=0.01, precision="bf16").to(dev).train()
eng = engine_of(m)
def eager():
    m.dropout_calls = 5
    out3, _ = eng.forward_backward(*a); torch.cuda.synchronize()
    return eng.flat.clone(), out3.clone()
e = [eager() for _ in range(4)]
words = torch.zeros(8, dtype=torch.int32, device=dev)
eng.step_words = words
m.dropout_calls = 5; m._packed_sig = None
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    with O.stream_scope():
        out3g, _ = eng.forward_backward(*a)
eng.step_words = None
def graphed():
    with O.stream_scope():
        O.store_words(words, [0, 0, 0, 2 * 6])
    g.replay(); torch.cuda.synchronize()
    return eng.flat.clone(), out3g.clone()
gr = [graphed() for _ in range(4)]
ref = e[0][0]
def rel(x): return float((x - ref).double().norm() / ref.double().norm())
print("eager vs eager:", [f"{rel(x[0]):.2e}" for x in e[1:]], "loss", [float(x[1][0]) for x in e])
print("graph vs eager:", [f"{rel(x[0]):.2e}" for x in gr], "loss", [float(x[1][0]) for x in gr])
# which parameters carry the difference
off = eng.offsets
d = (gr[0][0] - ref)
rows = sorted(((float(d[s:t].double().norm() / (ref[s:t].double().norm() + 1e-30)), n) for n, (s, t) in off.items()), reverse=True)[:6]
print("graph vs eager by parameter:", [(f"{r:.1e}", n) for r, n in rows])
d = (e[1][0] - ref)
rows = sorted(((float(d[s:t].double().norm() / (ref[s:t].double().norm() + 1e-30)), n) for n, (s, t) in off.items()), reverse=True)[:6]
print("eager vs eager by parameter:", [(f"{r:.1e}", n) for r, n in rows])
