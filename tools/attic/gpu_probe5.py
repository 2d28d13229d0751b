"""epilogue-bound launches: prenet-like Linear (K=128) at mel length, with / without the two outputs"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from efficient_tts_amd import lib as L, ops as P
dev = torch.device("cuda:0"); L.load(); L.require_device()
B, T, C, split = 64, 800, 512, 1
rs = P.Rows(B, T)
a = P.Plane.for_rows(rs, 128, split, dev)
x = torch.randn(B, T, 128, device=dev)
P.pack_rows(x, None, a, rs)
pw = P.PackedWeight(C, 128, 1, split, dev); pw.pack((torch.randn(C, 128, 1, device=dev) * 0.02).contiguous())
bias = torch.randn(C, device=dev)
gap = torch.zeros(rs.rows, device=dev); P.row_masks(torch.full((B,), T, dtype=torch.int32, device=dev), rs, gap, None)
out = P.F32Rows(rs, C, dev); outp = P.Plane.for_rows(rs, C, split, dev)
xf = P.F32Rows(rs, C, dev)
def run(f32=True, plane=True, resid=False):
    P.gemm(a=a, b_ptr=pw.ptr, ldb=pw.ld, b_tap_stride=pw.tap_stride, taps=1, m=rs.rows, n=C, bias=bias,
           rowmask_ptr=gap.data_ptr(), out_f32_ptr=out.ptr if f32 else None, ldo=C, out_plane=outp if plane else None,
           resid_ptr=xf.ptr if resid else None, ldr=C)
def timeit(fn, iters=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
with P.stream_scope():
    for f32, plane, resid in ((True, True, False), (True, False, False), (False, True, False), (True, True, True)):
        t = timeit(lambda: run(f32, plane, resid))
        mb = rs.rows * C * (4 * f32 + 2 * plane + 4 * resid) / 1e6 + rs.rows * 128 * 2 / 1e6
        print(f"f32={f32} plane={plane} resid={resid}: {t:.1f} us  {mb:.0f} MB  {mb / t / 1e3 * 1e3:.2f} GB/ms = {mb/t/1e6*1e6:.0f} ... {mb*1e6/(t*1e-6)/1e12:.2f} TB/s")
    # reference: plain copy bandwidth of torch
    src = torch.empty(rs.rows * C, device=dev); dst = torch.empty_like(src)
    t = timeit(lambda: dst.copy_(src))
    print(f"torch copy {src.numel()*4/1e6:.0f} MB: {t:.1f} us  {2*src.numel()*4/(t*1e-6)/1e12:.2f} TB/s (r+w)")
    t = timeit(lambda: dst.fill_(1.0))
    print(f"torch fill {src.numel()*4/1e6:.0f} MB: {t:.1f} us  {src.numel()*4/(t*1e-6)/1e12:.2f} TB/s (w)")
