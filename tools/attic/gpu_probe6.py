"""k5 conv at B=64: cost of the epilogue streams (residual read / fp32 write / plane write)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from efficient_tts_amd import lib as L, ops as P
dev = torch.device("cuda:0"); L.load(); L.require_device()
B, T, C, split = int(os.environ.get("PB", "64")), 800, 512, int(os.environ.get("PSPLIT", "1"))
rs = P.Rows(B, T)
a = P.Plane.for_rows(rs, C, split, dev)
x = torch.randn(B, T, C, device=dev)
xf = P.F32Rows(rs, C, dev); xf.view().copy_(x)
P.pack_rows(x, None, a, rs)
pw = P.PackedWeight(C, C, 5, split, dev); pw.pack((torch.randn(C, C, 5, device=dev) * 0.02).contiguous())
bias = torch.randn(C, device=dev)
gap = torch.zeros(rs.rows, device=dev); P.row_masks(torch.full((B,), T, dtype=torch.int32, device=dev), rs, gap, None)
out = P.F32Rows(rs, C, dev); outp = P.Plane.for_rows(rs, C, split, dev)
def run(resid, f32, plane):
    P.gemm(a=a, b_ptr=pw.ptr, ldb=pw.ld, b_tap_stride=pw.tap_stride, taps=5, m=rs.rows, n=C, act=L.ACT_LEAKY, slope=0.1, bias=bias,
           resid_ptr=xf.ptr if resid else None, ldr=C, rowmask_ptr=gap.data_ptr(), out_f32_ptr=out.ptr if f32 else None, ldo=C,
           out_plane=outp if plane else None)
def timeit(fn, iters=100):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
with P.stream_scope():
    for resid, f32, plane in ((1, 1, 1), (0, 1, 1), (1, 0, 1), (1, 1, 0), (0, 0, 1), (0, 1, 0)):
        print(f"resid={resid} f32={f32} plane={plane}: {timeit(lambda: run(resid, f32, plane)):.1f} us", flush=True)
