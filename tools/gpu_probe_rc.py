"""efts_resconv5 (persistent 8-wave k5 residual layer on hi/lo planes) vs efts_gemm: bit equality and time.
  PSHAPES: "BxT,..."; PSPLIT: 1 | 2; RCK: kernel choice (efts_resconv5_args.kernel); PPLAN: explicit tile schedule, classes of tile heights in half units, e.g. "7,6|6,7" """
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficient_tts_amd import lib as L, ops as P
dev = torch.device("cuda:0"); L.load(); L.require_device()
P.RC_KERNEL = int(os.environ.get("RCK", "0"))     # RCK: 0 / 1 the 8-wave kernel (default), 2 the one-wave-per-SIMD kernel where it applies
C = 512
def bf16_split(x):
    hi = x.to(torch.bfloat16); lo = (x - hi.float()).to(torch.bfloat16)
    return hi, lo
def where(a, b, name):
    d = (a.float() - b.float()).abs()
    bad = (d > 0).any(dim=1).nonzero().flatten()
    if bad.numel():
        r0 = bad[0].item()
        cols = (d[r0] > 0).nonzero().flatten()
        print(f"    {name}: {bad.numel()} bad rows of {a.shape[0]} (buffer rows, 8 guard rows first): first {bad[:6].tolist()} last {bad[-3:].tolist()};"
              f" row {r0}: {cols.numel()} bad cols, first {cols[:4].tolist()}, got {a[r0, cols[0]].item():.5f} want {b[r0, cols[0]].item():.5f}", flush=True)
def timeit(fn, iters=int(os.environ.get("PLOOP", "50"))):        # PLOOP: launches per timing sample (long loops for tools/gpu_power_trace.sh)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
def case(B, T, split, check=True, time=True):
    torch.manual_seed(B * 1000 + T)
    rs = P.Rows(B, T)
    x = torch.randn(B, T, C, device=dev)
    hi, lo = bf16_split(x)
    x16 = hi.float() + lo.float()                      # what the planes can represent
    hi, lo = bf16_split(x16)                           # (re-split: bf16(x16) may differ from bf16(x) at rounding ties)
    a = P.Plane.for_rows(rs, C, split, dev)
    P.pack_rows(x16, None, a, rs)                      # split 2: hi|lo chunks of x16 (exact); split 1: hi
    a_lo = None
    if split == 1:
        a_lo = P.Plane.for_rows(rs, C, 1, dev)
        P.pack_rows((x16 - hi.float()).contiguous(), None, a_lo, rs)
    xf = P.F32Rows(rs, C, dev); xf.view().copy_(x16)
    pw = P.PackedWeight(C, C, 5, split, dev); pw.pack((torch.randn(C, C, 5, device=dev) * 0.02).contiguous())
    bias = torch.randn(C, device=dev)
    lens = torch.randint(max(1, T // 2), T + 1, (B,), dtype=torch.int32, device=dev); lens[0] = T
    gap = torch.zeros(rs.rows, device=dev); P.row_masks(lens, rs, gap, None)
    # reference: efts_gemm with the fp32 residual
    o_ref, p_ref = P.F32Rows(rs, C, dev), P.Plane.for_rows(rs, C, 2, dev)
    def ref():
        P.gemm(a=a, b_ptr=pw.ptr, ldb=pw.ld, b_tap_stride=pw.tap_stride, taps=5, m=rs.rows, n=C, act=L.ACT_LEAKY, slope=0.1, bias=bias,
               resid_ptr=xf.ptr, ldr=C, rowmask_ptr=gap.data_ptr(), out_f32_ptr=o_ref.ptr, ldo=C, out_plane=p_ref)
    outs = {}
    def mk(mode):
        o = P.F32Rows(rs, C, dev)
        if mode == "f32in_split2out":
            y = P.Plane.for_rows(rs, C, 2, dev)
            return o, y, None, lambda: P.resconv5(x=a, x_f32_ptr=xf.ptr, ldr=C, w=pw, m=rs.rows, n=C, bias=bias, rowmask_ptr=gap.data_ptr(),
                                                  y_f32_ptr=o.ptr, ldo=C, y=y)
        if mode == "planes":
            ys = 2 if split == 2 else 1
            y = P.Plane.for_rows(rs, C, ys, dev)
            yl = P.Plane.for_rows(rs, C, 1, dev) if ys == 1 else None
            plan = None
            if os.environ.get("PPLAN"):
                plan = P.make_plan(rs.rows, [[int(v) for v in cl.split(",")] for cl in os.environ["PPLAN"].split("|")])
            return o, y, yl, lambda: P.resconv5(x=a, x_lo=a_lo, w=pw, m=rs.rows, n=C, bias=bias, rowmask_ptr=gap.data_ptr(), y=y, y_lo=yl, plan=plan)
    with P.stream_scope():
        ref(); torch.cuda.synchronize()
        ok = True
        for mode in os.environ.get("PMODES", "f32in_split2out,planes").split(","):
            o, y, yl, fn = mk(mode)
            fn(); torch.cuda.synchronize()
            if check:
                if mode == "f32in_split2out":
                    e1 = torch.equal(o.buf, o_ref.buf); e2 = torch.equal(y.buf, p_ref.buf)
                    print(f"  B={B} T={T} split={split} {mode}: f32 equal {e1} (max diff {(o.buf - o_ref.buf).abs().max().item():.3e}), plane equal {e2}", flush=True)
                    ok &= e1 and e2
                    if not e1: where(o.buf, o_ref.buf, "f32")
                else:
                    # hi/lo of the reference fp32 output
                    if split == 2:
                        e = torch.equal(y.buf, p_ref.buf)
                        print(f"  B={B} T={T} split=2 planes: hi|lo plane equal {e}", flush=True)
                    else:
                        rh, rl = bf16_split(o_ref.buf)
                        yh = y.buf.view(torch.bfloat16).view(rs.alloc, -1)[:, :C]; ylv = yl.buf.view(torch.bfloat16).view(rs.alloc, -1)[:, :C]
                        e = torch.equal(yh, rh) and torch.equal(ylv, rl)
                        print(f"  B={B} T={T} split=1 planes: hi equal {torch.equal(yh, rh)}, lo equal {torch.equal(ylv, rl)}", flush=True)
                        if not e: where(yh, rh, "hi"); where(ylv, rl, "lo")
                    ok &= e
            if time:
                ts = [timeit(fn) for _ in range(3)]
                print(f"  B={B} T={T} split={split} {mode}: " + " ".join(f"{t:.1f}" for t in ts) + " us", flush=True)
        if time and os.environ.get("PREF", "1") == "1":
            ts = [timeit(ref) for _ in range(3)]
            print(f"  B={B} T={T} split={split} efts_gemm: " + " ".join(f"{t:.1f}" for t in ts) + " us", flush=True)
    return ok
if __name__ == "__main__":
    split = int(os.environ.get("PSPLIT", "0"))
    shapes = os.environ.get("PSHAPES", "3x37,5x300,16x800,32x800,64x800")
    allok = True
    for sp in ((1, 2) if split == 0 else (split,)):
        for sh in shapes.split(","):
            B, T = map(int, sh.split("x"))
            allok &= case(B, T, sp, check=os.environ.get("PCHECK", "1") == "1", time=B * T >= 4000)
    print("ALL EQUAL" if allok else "MISMATCH", flush=True)
