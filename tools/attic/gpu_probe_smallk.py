"""Isolated launch times of the short-K / few-step efts_gemm launches on the mel chain of the B=64 forward (prenet, q.k^T,
alpha'.V, mel head), by tiling and by output streams (timing only, random operands).  PB / PT1 / PT2: shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from efficient_tts_amd import lib as L, ops as P
dev = torch.device("cuda:0"); L.load(); L.require_device()
C = 512
B, T1, T2 = int(os.environ.get("PB", 64)), int(os.environ.get("PT1", 128)), int(os.environ.get("PT2", 800))


def timeit(fn, iters=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def rnd_plane(pl):
    pl.buf.copy_(torch.randn(pl.buf.numel() // 2, device=dev).mul_(0.5).to(torch.bfloat16).view(torch.uint8).view(pl.buf.shape))
    return pl


rs1, rs2 = P.Rows(B, T1), P.Rows(B, T2)
gap2 = torch.zeros(rs2.rows, device=dev); len2 = torch.zeros(rs2.rows, device=dev)
P.row_masks(torch.full((B,), T2, dtype=torch.int32, device=dev), rs2, gap2, len2)
TIL = dict(auto=None, generic=P.L.TILING_GENERIC, narrow=P.L.TILING_NARROW)


def run(name, variants):
    for vname, fn in variants:
        try:
            t = timeit(fn)
            print(f"  {name:8s} {vname:44s} {t:7.1f} us", flush=True)
        except Exception as e:
            print(f"  {name:8s} {vname:44s} failed: {str(e)[:80]}", flush=True)


with P.stream_scope():
    for split in (1, 2):
        print(f"split {split}  B={B} T1={T1} T2={T2}", flush=True)
        # ---- prenet: [rows, 80] x [80, 512], leaky, gap mask
        mel_in = rnd_plane(P.Plane.for_rows(rs2, 80, split, dev))
        wp = P.PackedWeight(C, 80, 1, split, dev); wp.pack((torch.randn(C, 80, device=dev) * 0.1).contiguous())
        bias = torch.randn(C, device=dev)
        o_f = P.F32Rows(rs2, C, dev)
        o_p = P.Plane.for_rows(rs2, C, split, dev)
        o_l = P.Plane.for_rows(rs2, C, 1, dev)
        def prenet(til, f32, plane, lo):
            return lambda: P.gemm(a=mel_in, b_ptr=wp.ptr, ldb=wp.ld, m=rs2.rows, n=C, act=L.ACT_LEAKY, slope=0.1, bias=bias,
                                  rowmask_ptr=gap2.data_ptr(), out_f32_ptr=o_f.ptr if f32 else None, ldo=C,
                                  out_plane=o_p if plane else None, out_plane_lo=o_l if lo else None, tiling=TIL[til])
        v = [("generic plane" + (" + lo" if split == 1 else ""), prenet("generic", False, True, split == 1)),
             ("generic plane only", prenet("generic", False, True, False)),
             ("narrow  plane only", prenet("narrow", False, True, False)),
             ("generic f32 only", prenet("generic", True, False, False)),
             ("narrow  f32 only", prenet("narrow", True, False, False)),
             ("generic f32 + plane", prenet("generic", True, True, False)),
             ("narrow  f32 + plane", prenet("narrow", True, True, False))]
        run("prenet", v)
        # ---- alpha'.V: per item [T2, T1] x [T1, 512]
        ra = rnd_plane(P.Plane.for_rows(rs2, T1, 2, dev))
        vt = rnd_plane(P.Plane(B * C, T1, 2, dev))
        def pv(til, f32, plane, lo):
            return lambda: P.gemm(a=ra, b_ptr=vt.ptr, ldb=vt.ld, m=rs2.T, n=C, batch=B, a_batch_stride=rs2.Tp * ra.ld,
                                  b_batch_stride=C * vt.ld, rowmask_ptr=len2.data_ptr(), rowmask_batch_stride=rs2.Tp,
                                  out_f32_ptr=o_f.ptr if f32 else None, ldo=C, out_batch_stride=rs2.Tp * C,
                                  out_plane=o_p if plane else None, outb_batch_stride=rs2.Tp * o_p.ld,
                                  out_plane_lo=o_l if lo else None, tiling=TIL[til])
        v = [("generic plane" + (" + lo" if split == 1 else ""), pv("generic", False, True, split == 1)),
             ("generic plane only", pv("generic", False, True, False)),
             ("narrow  plane only", pv("narrow", False, True, False)),
             ("generic f32 + plane", pv("generic", True, True, False)),
             ("narrow  f32 + plane", pv("narrow", True, True, False))]
        run("alphaV", v)
        if split == 1:
            # ---- q.k^T: per item [T2, 512] x [512, T1] on split-2 planes, fp32 scores
            q = rnd_plane(P.Plane.for_rows(rs2, C, 2, dev))
            kp = rnd_plane(P.Plane.for_rows(rs1, C, 2, dev))
            sc = torch.empty(B, T2, T1, device=dev)
            def qk(til):
                return lambda: P.gemm(a=q, b_ptr=kp.ptr, ldb=kp.ld, m=T2, n=T1, batch=B, a_batch_stride=rs2.Tp * q.ld,
                                      b_batch_stride=rs1.Tp * kp.ld, alpha=0.044, out_f32_ptr=sc.data_ptr(), ldo=T1,
                                      out_batch_stride=T2 * T1, tiling=TIL[til])
            run("qk", [("generic", qk("generic")), ("narrow", qk("narrow")), ("auto", qk("auto"))])
        # ---- mel head: [rows, 512] x [512, 80]
        d_p = rnd_plane(P.Plane.for_rows(rs2, C, split, dev))
        wh = P.PackedWeight(80, C, 1, split, dev); wh.pack((torch.randn(80, C, device=dev) * 0.05).contiguous())
        mel = P.F32Rows(rs2, 80, dev)
        hb = torch.randn(80, device=dev)
        def head(til):
            return lambda: P.gemm(a=d_p, b_ptr=wh.ptr, ldb=wh.ld, m=rs2.rows, n=80, bias=hb, rowmask_ptr=len2.data_ptr(),
                                  out_f32_ptr=mel.ptr, ldo=80, tiling=TIL[til])
        run("head", [("generic", head("generic")), ("narrow", head("narrow")), ("auto", head("auto"))])
