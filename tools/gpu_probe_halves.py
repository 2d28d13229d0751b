"""Does running the B=64 forward as two concurrent B=32 halves (two streams, one hipGraph) beat the single B=64 pass?"""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficient_tts_amd import EfficientTTSCNN
dev = torch.device("cuda:0")
prec = os.environ.get("PREC", "bf16")
torch.manual_seed(0)
def mk():
    torch.manual_seed(0)
    return EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, use_weighted_masking=False, sigma=0.01, precision=prec).to(dev).eval()
def synth(B, T1=128, T2=800):
    g = torch.Generator(device="cpu").manual_seed(1234)
    return (torch.randint(0, 76, (B, T1), generator=g).to(dev), torch.full((B,), T1).to(dev),
            torch.randn(B, T2, 80, generator=g).to(dev), torch.full((B,), T2).to(dev))
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def graphed(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    return g.replay
for nparts in (1, 2, 4):
    B = 64 // nparts
    models = [mk() for _ in range(nparts)]
    if os.environ.get("NOSIDE"):
        for m_ in models:                                   # no nested fork inside the capture: the other parts provide the overlap
            object.__setattr__(m_, "_side_stream", lambda d: torch.cuda.current_stream(d))
    args = [synth(B) for _ in range(nparts)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(nparts)]
    def step():
        with torch.no_grad():
            cur = torch.cuda.current_stream(dev)
            if nparts == 1:
                models[0](*args[0])
            else:
                for m, a, st in zip(models, args, streams):
                    st.wait_stream(cur)
                    with torch.cuda.stream(st):
                        m(*a)
                for st in streams:
                    cur.wait_stream(st)
    ms = timeit(step, 50)
    print(f"{prec}: {nparts} x B={B} concurrently, eager: {ms:.3f} ms  -> {64*800/ms/1e3:.2f} M frames/s", flush=True)
    if nparts == 1:
        ms = timeit(graphed(step))
        print(f"{prec}: {nparts} x B={B}, hipGraph: {ms:.3f} ms  -> {64*800/ms/1e3:.2f} M frames/s", flush=True)
