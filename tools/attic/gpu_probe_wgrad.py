"""TN wgrad kernel vs a float64 reference and vs the transposed-plane path; timing"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from efficient_tts_amd import lib as L, ops as P
dev = torch.device("cuda:0"); lib = L.load(); L.require_device()
def run(B, T, C=512, S=6, check=True, iters=20):
    rs = P.Rows(B, T)
    x = torch.randn(B, T, C, device=dev); dz = torch.randn(B, T, C, device=dev) * 0.1
    xp = P.Plane.for_rows(rs, C, 1, dev); zp = P.Plane.for_rows(rs, C, 1, dev)
    P.pack_rows(x, None, xp, rs); P.pack_rows(dz, None, zp, rs)
    part = torch.empty(5, S, C, C, device=dev)
    def go():
        L.check(lib.efts_wgrad_tn(zp.ptr, zp.ld, xp.ptr, xp.ld, part.data_ptr(), rs.rows, C, C, 5, S, 1, P._stream()), "wgrad_tn")
    go(); torch.cuda.synchronize()
    dw = part.sum(1)                               # [5][co][ci]
    if check:
        xb = x.to(torch.bfloat16).double(); zb = dz.to(torch.bfloat16).double()
        xpad = torch.nn.functional.pad(xb, (0, 0, 2, 2))
        ref = torch.stack([torch.einsum("bto,bti->oi", zb, xpad[:, k:k + T]) for k in range(5)])
        err = (dw.double() - ref).abs().max().item(); scale = ref.abs().max().item()
        print(f"B={B} T={T} S={S}: max err {err:.3e} (ref max {scale:.3e})", flush=True)
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): go()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"   {us:.1f} us  {2.0*B*T*C*C*5/us/1e6:.0f} TFLOP/s", flush=True)
with P.stream_scope():
    run(2, 100, S=2)
    run(3, 333, S=5)
    run(32, 800, S=6)
    run(32, 800, S=8, check=False)
    run(32, 800, S=16, check=False)
