"""host enqueue time vs device time of the graphed forward (one chain / two pipelines)"""
import sys, time, torch
sys.path.insert(0, ".")
from efficient_tts_amd import EfficientTTSCNN
dev = torch.device("cuda:0")
B, T1, T2 = 64, 128, 800
m = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, sigma=0.01, precision="bf16").to(dev).eval()
text = torch.randint(0, 76, (B, T1), device=dev); mel = torch.randn(B, T2, 80, device=dev)
tl, ml = torch.full((B,), T1, device=dev), torch.full((B,), T2, device=dev)
for npipe in (1, 2):
    m.pipelines = npipe
    with torch.no_grad():
        for _ in range(6): m(text, tl, mel, ml)
        torch.cuda.synchronize()
        for rep in range(2):
            t0 = time.perf_counter()
            for _ in range(40): m(text, tl, mel, ml)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print(f"pipelines {npipe}: host enqueue {1e3*(t1-t0)/40:.3f} ms/step, total {1e3*(t2-t0)/40:.3f} ms/step")
        # host time of the pieces of one call
        ent = list(m._graph_cache.entries.values())[-1]
        gs = (ent.parts or []) + [ent.graph]
        for g in gs:
            torch.cuda.synchronize()
            t0 = time.perf_counter(); g.replay(); t1 = time.perf_counter()
            torch.cuda.synchronize()
            print(f"   one graph replay: host {1e6*(t1-t0):.0f} us, to completion {1e6*(time.perf_counter()-t0):.0f} us")
