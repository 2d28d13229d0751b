"""one-tap efts_gemm launches of the forward, each alone in a loop: us per launch against the number of K chunks (slope = per step, intercept = fixed)
python tools/gpu_probe_gemm1.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficient_tts_amd import lib as L, ops as P

dev = torch.device("cuda:0")
L.load()


def timeit(fn, n=200, warm=20):
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


def linear(rows_B, T, cin, cout, split, out="plane2", tiling=None):
    rs = P.Rows(rows_B, T)
    a = P.Plane.for_rows(rs, cin, split, dev)
    a.buf.view(torch.int16).random_(-20000, 20000)
    pw = P.PackedWeight(cout, cin, 1, split, dev)
    pw.pack((torch.randn(cout, cin, 1, device=dev) * 0.05).contiguous())
    bias = torch.randn(cout, device=dev)
    gap = torch.ones(rs.rows, device=dev)
    of = P.F32Rows(rs, cout, dev)
    op = P.Plane.for_rows(rs, cout, 2 if out == "plane2" else split, dev)
    res = {}
    for nchunk in sorted({1, 2, 4, a.nchunk // 2, a.nchunk}):
        def fn():
            P.gemm(a=a, b_ptr=pw.ptr, ldb=pw.ld, m=rs.rows, n=cout, bias=bias, rowmask_ptr=gap.data_ptr(), nchunk=nchunk,
                   out_f32_ptr=of.ptr if out != "plane2" else None, ldo=cout, out_plane=op if out != "f32" else None, tiling=tiling)
        res[nchunk] = round(timeit(fn), 1)
    return res


def qk(B, T2, T1, C):
    rs2, rs1 = P.Rows(B, T2), P.Rows(B, T1)
    q = P.Plane.for_rows(rs2, C, 2, dev)
    k = P.Plane.for_rows(rs1, C, 2, dev)
    q.buf.view(torch.int16).random_(-20000, 20000)
    k.buf.view(torch.int16).random_(-20000, 20000)
    sidx = torch.zeros(B, T2, device=dev)
    tl = torch.full((B,), T1, dtype=torch.int32, device=dev)
    ml = torch.full((B,), T2, dtype=torch.int32, device=dev)
    res = {}
    for nchunk in (1, 2, 4, 8, 16):
        def fn():
            P.gemm(a=q, b_ptr=k.ptr, ldb=k.ld, m=T2, n=T1, batch=B, a_batch_stride=rs2.Tp * q.ld, b_batch_stride=rs1.Tp * k.ld, alpha=0.044,
                   soft_index=sidx, key_len=tl, query_len=ml, nchunk=nchunk)
        res[nchunk] = round(timeit(fn), 1)
    return res


print("key projection  (8320 rows, 512 -> 512, bf16 planes in, split-2 plane out), us by K chunks:", linear(64, 128, 512, 512, 1))
print("key projection  (same, split-2 planes in)                                                :", linear(64, 128, 512, 512, 2))
print("value projection (fp32 + plane out)                                                      :", linear(64, 128, 512, 512, 1, out="both"))
print("mel head        (51328 rows, 512 -> 80, fp32 out)                                        :", linear(64, 800, 512, 80, 1, out="f32"))
print("mel head, generic tiling forced                                                          :", linear(64, 800, 512, 80, 1, out="f32", tiling=L.TILING_GENERIC))
print("q.k^T + soft index (64 x 800 x 128, K 512 hi/lo)                                         :", qk(64, 800, 128, 512))
