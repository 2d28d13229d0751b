"""Throughput of P independent B=64/P forward pipelines, each driven eagerly by its own host thread on its own stream"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from efficient_tts_amd import EfficientTTSCNN
dev = torch.device("cuda:0")
prec = os.environ.get("PREC", "bf16")
def mk():
    torch.manual_seed(0)
    return EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, use_weighted_masking=False, sigma=0.01, precision=prec).to(dev).eval()
def synth(B, T1=128, T2=800):
    g = torch.Generator(device="cpu").manual_seed(1234)
    return (torch.randint(0, 76, (B, T1), generator=g).to(dev), torch.full((B,), T1).to(dev),
            torch.randn(B, T2, 80, generator=g).to(dev), torch.full((B,), T2).to(dev))
N = 60
for nparts in (1, 2):
    B = 64 // nparts
    models = [mk() for _ in range(nparts)]
    args = [synth(B) for _ in range(nparts)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(nparts)]
    def worker(i, n):
        with torch.no_grad(), torch.cuda.stream(streams[i]):
            for _ in range(n):
                models[i](*args[i])
    for i in range(nparts): worker(i, 3)
    torch.cuda.synchronize()
    ths = [threading.Thread(target=worker, args=(i, N)) for i in range(nparts)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N
    print(f"{prec}: {nparts} thread(s) x B={B}: {dt*1e3:.3f} ms per 64 items -> {64*800/dt/1e6:.2f} M frames/s", flush=True)
