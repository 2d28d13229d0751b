"""generalised contraction: taps 3/7/11 with dilation, narrow channels, activated plane, tanh -- against torch conv1d"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from efficient_tts_amd import lib as L, ops as P
dev = torch.device("cuda:0"); L.load(); L.require_device()
G = 64
def run(T, cin, cout, taps, dil, split, act=L.ACT_NONE, plane_act=False, resid=False):
    g = torch.Generator().manual_seed(T + cin + taps)
    x = torch.randn(1, T, cin, generator=g).to(dev)
    w = (torch.randn(cout, cin, taps, generator=g) / (cin * taps) ** 0.5).to(dev)
    b = torch.randn(cout, generator=g).to(dev)
    rows = T
    alloc = G + P.roundup(rows, 128) + 192
    a = P.Plane(alloc, cin, split, dev, guard_lo=G)
    xs = torch.zeros(alloc, cin, device=dev); xs[G:G + T] = x[0]
    L.check(L.load().efts_pack_rows(xs[G:].data_ptr(), None, a.ptr, a.ld, 1, T, T, cin, a.nchunk * P.chunk_k(split), split, P._stream()), "pack")
    pw = P.PackedWeight(cout, cin, taps, split, dev); pw.pack(w.contiguous())
    out = torch.zeros(alloc, cout, device=dev)
    outp = P.Plane(alloc, cout, split, dev, guard_lo=G)
    rx = torch.randn(alloc, cout, device=dev) if resid else None
    P.gemm(a=a, b_ptr=pw.ptr, ldb=pw.ld, b_tap_stride=pw.tap_stride, taps=taps, m=rows, n=cout, act=act, slope=0.1, bias=b,
           resid_ptr=None if rx is None else rx[G:].data_ptr(), ldr=cout, out_f32_ptr=out[G:].data_ptr(), ldo=cout, out_plane=outp,
           dilation=dil, plane_act=plane_act, plane_slope=0.1)
    torch.cuda.synchronize()
    xr = x if split == 2 else x.to(torch.bfloat16).float()
    wr = w if split == 2 else w.to(torch.bfloat16).float()
    ref = F.conv1d(xr.transpose(1, 2).double(), wr.double(), b.double(), padding=(taps - 1) // 2 * dil, dilation=dil).transpose(1, 2)[0]
    if act == L.ACT_LEAKY: ref = F.leaky_relu(ref, 0.1)
    if act == L.ACT_TANH: ref = torch.tanh(ref)
    if resid: ref = ref + rx[G:G + T].double()
    err = (out[G:G + T].double() - ref).abs().max().item()
    # plane check (hi part only): compare bf16(act?(out)) decoded
    pl = outp.buf[G:G + T].view(torch.int16)            # [T][ld/2]
    ck = P.chunk_k(split)
    got = torch.stack([pl[:, (c // ck) * 64 + (c % ck)] for c in range(cout)], 1).view(torch.bfloat16).float()
    want = out[G:G + T]
    if plane_act: want = F.leaky_relu(want, 0.1)
    perr = (got - want.to(torch.bfloat16).float()).abs().max().item()
    tail = out[G + T:G + T + 64].abs().max().item()
    print(f"T={T} cin={cin} cout={cout} taps={taps} dil={dil} split={split} act={act} pa={int(plane_act)} res={int(resid)}: err {err:.2e} plane {perr:.2e} tail {tail:.1e} (ref max {ref.abs().max().item():.2f})", flush=True)
with P.stream_scope():
    for split in (2, 1):
        run(300, 512, 512, 5, 1, split, act=L.ACT_LEAKY, resid=True)
        run(300, 256, 256, 3, 1, split)
        run(300, 256, 256, 3, 3, split, plane_act=True)
        run(500, 128, 128, 7, 5, split, resid=True, plane_act=True)
        run(700, 64, 64, 11, 5, split, resid=True)
        run(700, 32, 32, 11, 3, split, plane_act=True)
        run(130, 80, 512, 7, 1, split)
        run(260, 32, 1, 7, 1, split, act=L.ACT_TANH)
        run(100, 512, 2048, 3, 1, split)
