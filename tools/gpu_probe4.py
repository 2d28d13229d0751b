import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficient_tts_amd import lib as L, ops as P
dev = torch.device("cuda:0"); L.load(); L.require_device()
B, T, C, split = 64, 800, 512, 1
rs = P.Rows(B, T)
a = P.Plane.for_rows(rs, C, split, dev)
x = torch.randn(B, T, C, device=dev)
xf = P.F32Rows(rs, C, dev); xf.view().copy_(x)
P.pack_rows(x, None, a, rs)
pw = P.PackedWeight(C, C, 5, split, dev); pw.pack((torch.randn(C, C, 5, device=dev) * 0.02).contiguous())
bias = torch.randn(C, device=dev)
gap = torch.zeros(rs.rows, device=dev); P.row_masks(torch.full((B,), T, dtype=torch.int32, device=dev), rs, gap, None)
out = P.F32Rows(rs, C, dev); outp = P.Plane.for_rows(rs, C, split, dev)
def run(r0, r1):
    P.gemm(a=a, a_ptr=a.ptr + r0 * a.ld, b_ptr=pw.ptr, ldb=pw.ld, b_tap_stride=pw.tap_stride, taps=5, m=r1 - r0, n=C, act=L.ACT_LEAKY, slope=0.1,
           bias=bias, resid_ptr=xf.ptr + r0 * C * 4, ldr=C, rowmask_ptr=gap.data_ptr() + r0 * 4, out_f32_ptr=out.ptr + r0 * C * 4, ldo=C,
           out_plane=outp, out_plane_ptr=outp.ptr + r0 * outp.ld)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); 
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
M = rs.rows
def one(): run(0, M)
def two_streams(nsplit=2):
    main = torch.cuda.current_stream()
    streams = [s1, s2]
    for i, st in enumerate(streams):
        st.wait_stream(main)
        with P.on_stream(st):
            lo = (M * i // 2) // 128 * 128; hi = M if i == 1 else (M * (i + 1) // 2) // 128 * 128
            run(lo, hi)
    for st in streams: main.wait_stream(st)
def seq_halves():
    h = (M // 2) // 128 * 128
    run(0, h); run(h, M)
with P.stream_scope():
    print("one launch      :", round(timeit(one), 1), "us")
print("two streams     :", round(timeit(two_streams), 1), "us")
with P.stream_scope():
    print("two seq halves  :", round(timeit(seq_halves), 1), "us")
