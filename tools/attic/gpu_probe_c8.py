"""k5 conv at B=64 / 32: gemm_kernel vs conv5_kernel vs conv8_kernel (8 waves, 256 x 256 tile): equality and time"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from efficient_tts_amd import lib as L, ops as P
dev = torch.device("cuda:0"); L.load(); L.require_device()
B, T, C, split = int(os.environ.get("PB", "64")), int(os.environ.get("PT", "800")), 512, 1
torch.manual_seed(0)
rs = P.Rows(B, T)
a = P.Plane.for_rows(rs, C, split, dev)
x = torch.randn(B, T, C, device=dev)
xf = P.F32Rows(rs, C, dev); xf.view().copy_(x)
P.pack_rows(x, None, a, rs)
pw = P.PackedWeight(C, C, 5, split, dev); pw.pack((torch.randn(C, C, 5, device=dev) * 0.02).contiguous())
bias = torch.randn(C, device=dev)
gap = torch.zeros(rs.rows, device=dev); P.row_masks(torch.full((B,), T, dtype=torch.int32, device=dev), rs, gap, None)
def mk(): return P.F32Rows(rs, C, dev), P.Plane.for_rows(rs, C, split, dev)
def run(out, outp):
    P.gemm(a=a, b_ptr=pw.ptr, ldb=pw.ld, b_tap_stride=pw.tap_stride, taps=5, m=rs.rows, n=C, act=L.ACT_LEAKY, slope=0.1, bias=bias,
           resid_ptr=xf.ptr, ldr=C, rowmask_ptr=gap.data_ptr(), out_f32_ptr=out.ptr, ldo=C, out_plane=outp)
def timeit(fn, iters=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
res = {}
with P.stream_scope():
    for name, env in (("gemm", {"EFTS_CONV5": "0"}), ("conv5", {"EFTS_CONV5": "1"}), ("conv8", {"EFTS_CONV5": "1", "EFTS_CONV8": "1"})):
        for k in ("EFTS_CONV5", "EFTS_CONV8"): os.environ.pop(k, None)
        os.environ.update(env)
        o, op = mk()
        run(o, op); torch.cuda.synchronize()
        res[name] = (o.buf.clone(), op.buf.clone())
        ts = [timeit(lambda: run(o, op)) for _ in range(3)]
        print(name, " ".join(f"{t:.1f}" for t in ts), "us", flush=True)
    for name in ("conv5", "conv8"):
        d = (res[name][0] - res["gemm"][0]).abs().max().item()
        print(f"{name} vs gemm: f32 max diff {d:.3e}, plane equal {bool(torch.equal(res[name][1], res['gemm'][1]))}", flush=True)
