"""the prenet at 64 x 800 frames: efts_frame_linear against efts_pack_rows + efts_gemm (us, isolated loops)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficient_tts_amd import lib as L, ops as P
dev = torch.device("cuda:0"); L.load(); L.require_device()
B, T, C = int(os.environ.get("PB", 64)), 800, 512
def t(fn, n=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
rs = P.Rows(B, T)
x = torch.randn(B, T, 80, device=dev); bias = torch.randn(C, device=dev)
gap = torch.zeros(rs.rows, device=dev); P.row_masks(torch.full((B,), T, dtype=torch.int32, device=dev), rs, gap, None)
with P.stream_scope():
    for split in (1, 2):
        pw = P.PackedWeight(C, 80, 1, split, dev); pw.pack((torch.randn(C, 80, device=dev) * 0.1).contiguous())
        a = P.Plane.for_rows(rs, 80, split, dev); y = P.Plane.for_rows(rs, C, split, dev)
        yl = P.Plane.for_rows(rs, C, 1, dev) if split == 1 else None
        def old():
            P.pack_rows(x, None, a, rs)
            P.gemm(a=a, b_ptr=pw.ptr, ldb=pw.ld, m=rs.rows, n=C, act=L.ACT_LEAKY, slope=0.1, bias=bias, rowmask_ptr=gap.data_ptr(), out_plane=y, out_plane_lo=yl)
        new = lambda: P.frame_linear(x=x, w=pw, bias=bias, act=L.ACT_LEAKY, slope=0.1, rs=rs, y=y, y_lo=yl)
        print(f"split {split}: pack_rows + gemm {t(old):.1f} us   frame_linear {t(new):.1f} us", flush=True)
