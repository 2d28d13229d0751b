"""conv8 (one 8-wave workgroup per CU, 256 x 256 tile) at growing grids: per-tile time with 52 ... 408 workgroups.
Question: is the time outside the main loop (epilogue, prologue) the HBM burst of all CUs at once?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from efficient_tts_amd import lib as L, ops as P
dev = torch.device("cuda:0"); L.load(); L.require_device()
os.environ["EFTS_CONV5"] = "1"; os.environ["EFTS_CONV8"] = "1"
T, C, split = 800, 512, 1
def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for B in (2, 4, 8, 16, 24, 32, 40, 64):
    torch.manual_seed(0)
    rs = P.Rows(B, T)
    a = P.Plane.for_rows(rs, C, split, dev)
    x = torch.randn(B, T, C, device=dev)
    xf = P.F32Rows(rs, C, dev); xf.view().copy_(x)
    P.pack_rows(x, None, a, rs)
    pw = P.PackedWeight(C, C, 5, split, dev); pw.pack((torch.randn(C, C, 5, device=dev) * 0.02).contiguous())
    bias = torch.randn(C, device=dev)
    gap = torch.zeros(rs.rows, device=dev); P.row_masks(torch.full((B,), T, dtype=torch.int32, device=dev), rs, gap, None)
    o, op = P.F32Rows(rs, C, dev), P.Plane.for_rows(rs, C, split, dev)
    def run():
        P.gemm(a=a, b_ptr=pw.ptr, ldb=pw.ld, b_tap_stride=pw.tap_stride, taps=5, m=rs.rows, n=C, act=L.ACT_LEAKY, slope=0.1, bias=bias,
               resid_ptr=xf.ptr, ldr=C, rowmask_ptr=gap.data_ptr(), out_f32_ptr=o.ptr, ldo=C, out_plane=op)
    with P.stream_scope():
        ts = [timeit(run) for _ in range(3)]
    tiles = ((rs.rows + 251) // 252) * 2
    print(f"B={B} rows={rs.rows} tiles={tiles} ", " ".join(f"{t:.1f}" for t in ts), "us", flush=True)
