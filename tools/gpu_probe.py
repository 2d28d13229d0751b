#!/usr/bin/env python3
"""Quick on-GPU probe: time one decoder-shaped conv layer and the whole forward in both modes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficient_tts_amd import EfficientTTSCNN, lib as L, ops as P
from oracle import efts_oracle as O

dev = torch.device("cuda:0")
L.load(); L.require_device()

def time_layer(split, B=64, T=800, C=512, iters=20):
    rs = P.Rows(B, T)
    a = P.Plane.for_rows(rs, C, split, dev)
    x = torch.randn(B, T, C, device=dev)
    xf = P.F32Rows(rs, C, dev); xf.view().copy_(x)
    P.pack_rows(x, None, a, rs)
    pw = P.PackedWeight(C, C, 5, split, dev); pw.pack((torch.randn(C, C, 5, device=dev) * 0.02).contiguous())
    bias = torch.randn(C, device=dev)
    gap = torch.zeros(rs.rows, device=dev); P.row_masks(torch.full((B,), T, dtype=torch.int32, device=dev), rs, gap, None)
    out = P.F32Rows(rs, C, dev); outp = P.Plane.for_rows(rs, C, split, dev)
    def run():
        P.gemm(a=a, b_ptr=pw.ptr, ldb=pw.ld, b_tap_stride=pw.tap_stride, taps=5, m=rs.rows, n=C, act=L.ACT_LEAKY, slope=0.1,
               bias=bias, resid_ptr=xf.ptr, ldr=C, rowmask_ptr=gap.data_ptr(), out_f32_ptr=out.ptr, ldo=C, out_plane=outp)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * B * T * C * C * 5
    print(f"conv k5 C=512 B={B} T={T} split={split}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TFLOP/s algorithmic")

def time_forward(prec, B=64, T1=128, T2=800, iters=10):
    m = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, sigma=0.01, precision=prec)
    m.load_state_dict(O.fill_params()); m = m.to(dev).eval()
    g = torch.Generator().manual_seed(1234)
    text = torch.randint(0, 76, (B, T1), generator=g).to(dev); mel = torch.randn(B, T2, 80, generator=g).to(dev)
    tl = torch.full((B,), T1, dtype=torch.int64, device=dev); sl = torch.full((B,), T2, dtype=torch.int64, device=dev)
    with torch.no_grad():
        for _ in range(2): m(text, tl, mel, sl)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(iters): m(text, tl, mel, sl)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
    print(f"forward {prec} B={B}: {dt*1e3:.2f} ms  {B*T2/dt/1e6:.2f} M mel-frames/s  {21.43e9*B/dt/1e12:.1f} TFLOP/s")

for s in (1, 2): time_layer(s)
for s in (1, 2): time_layer(s, B=64, T=128)
for p in ("bf16", "bf16x3"): time_forward(p)
