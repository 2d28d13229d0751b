"""A/B of model switches inside one process, arms interleaved: python tools/gpu_ab_forward.py name=attr:val[,attr:val] ...
e.g.  python tools/gpu_ab_forward.py base= nosoft=fuse_soft_index:0 noexp=fuse_expand:0     (precisions bf16 and bf16x3, B=64; attributes = the fields of model.opt)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficient_tts_amd import EfficientTTSCNN  # noqa: E402


def timeit(fn, n=20, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def main():
    arms = []
    for a in sys.argv[1:]:
        name, _, spec = a.partition("=")
        kv = [x.split(":") for x in spec.split(",") if x]
        arms.append((name, [(k, int(v)) for k, v in kv]))
    B, T1, T2 = (int(x) for x in os.environ.get("SHAPE", "64,128,800").split(","))
    dev = torch.device("cuda:0")
    for prec in os.environ.get("PRECS", "bf16,bf16x3").split(","):
        m = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, sigma=0.01, precision=prec).to(dev).eval()
        defaults = {k: getattr(m, k) for _, kv in arms for k, _ in kv}
        text = torch.randint(0, 76, (B, T1), device=dev)
        mel = torch.randn(B, T2, 80, device=dev)
        tl, ml = torch.full((B,), T1, device=dev), torch.full((B,), T2, device=dev)
        res, ref = {}, {}
        for rep in range(3):
            for name, kv in arms:
                for k, v in defaults.items():
                    setattr(m, k, v)
                for k, v in kv:
                    setattr(m, k, type(defaults[k])(v))
                with torch.no_grad():
                    out = m(text, tl, mel, ml)
                    res.setdefault(name, []).append(timeit(lambda: m(text, tl, mel, ml)))
                ref.setdefault(name, out[4].clone())
        base = ref[arms[0][0]]
        for name, v in res.items():
            print(f"  forward {prec:7s} {name:14s}: " + " ".join(f"{x:8.1f}" for x in v) + f" us   mel max-abs vs first arm {float((ref[name] - base).abs().max()):.2e}")


if __name__ == "__main__":
    main()
