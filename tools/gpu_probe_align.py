"""GPU probe: the fused alignment kernels (efts_imv_align, efts_expand) against the chain they replace, isolated and inside the
forward (A/B in one process, arms interleaved).  python tools/gpu_probe_align.py [B T1 T2]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from efficient_tts_amd import EfficientTTSCNN, ops as P  # noqa: E402


def timeit(fn, n=50, warm=5):
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


def main():
    B, T1, T2 = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (64, 128, 800)
    dev = torch.device("cuda:0")
    C = 512
    g = torch.Generator().manual_seed(1)
    tl = torch.full((B,), T1, dtype=torch.int32, device=dev)
    ml = torch.full((B,), T2, dtype=torch.int32, device=dev)
    sidx = (torch.rand(B, T2, generator=g).cumsum(1) / T2 * T1 * 2).to(dev)
    imv, e, lde = torch.empty(B, T2, device=dev), torch.empty(B, T1, device=dev), torch.empty(B, T1, device=dev)
    rs1, rs2 = P.Rows(B, T1), P.Rows(B, T2)
    vf = P.F32Rows(rs1, C, dev)
    vf.view().copy_(torch.randn(B, T1, C, generator=g).to(dev))
    ra = torch.empty(B, T1, T2, device=dev)
    ra_p = P.Plane.for_rows(rs2, T1, 2, dev)
    vt = P.Plane(B * C, T1, 2, dev)
    lenm, gap = torch.zeros(rs2.rows, device=dev), torch.zeros(rs2.rows, device=dev)
    P.row_masks(ml, rs2, gap, lenm)

    def chain_align():
        P.imv_scan(sidx, tl, ml, imv, B, T2)
        P.aligned_positions(imv, tl, ml, 0.5, 1.0, e, lde, B, T1, T2)

    def fused_align():
        P.imv_align(sidx, tl, ml, 0.5, 1.0, True, imv, e, lde, B, T1, T2)

    print(f"B={B} T1={T1} T2={T2}")
    print(f"  imv_scan + aligned_pos + dur_target : {timeit(chain_align):8.1f} us")
    print(f"  efts_imv_align                      : {timeit(fused_align):8.1f} us")
    for split in (1, 2):
        y_p = P.Plane.for_rows(rs2, C, split, dev)
        y_l = P.Plane.for_rows(rs2, C, 1, dev) if split == 1 else None

        def chain_expand():
            P.reconst_alpha(e, tl, ml, 0.01, ra, ra_p, B, T1, T2, rs2.Tp)
            P.pack_vt(vf, vt, B, T1, rs1.Tp, C)
            P.gemm(a=ra_p, b_ptr=vt.ptr, ldb=vt.ld, m=T2, n=C, batch=B, a_batch_stride=rs2.Tp * ra_p.ld, b_batch_stride=C * vt.ld,
                   rowmask_ptr=lenm.data_ptr(), rowmask_batch_stride=rs2.Tp, out_plane=y_p, outb_batch_stride=rs2.Tp * y_p.ld,
                   out_plane_lo=y_l)

        def fused_expand():
            P.expand(e=e, tl=tl, ml=ml, sigma=0.01, v=vf, rs1=rs1, rs2=rs2, alpha_out=ra, y=y_p, y_lo=y_l)

        print(f"  out split {split}: reconst_alpha + pack_vt + gemm : {timeit(chain_expand):8.1f} us")
        print(f"  out split {split}: efts_expand                    : {timeit(fused_expand):8.1f} us")
        out_mb = rs2.rows * C * 4 / 1e6 + B * T1 * T2 * 4 / 1e6
        print(f"               ({out_mb:.0f} MB written -> {out_mb / timeit(fused_expand) * 1e-3 * 1e3:.2f} TB/s... "
              f"see the line above)")

    # the forward, A/B
    for prec in ("bf16", "bf16x3"):
        m = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, sigma=0.01, precision=prec).to(dev).eval()
        text = torch.randint(0, 76, (B, T1), device=dev)
        mel = torch.randn(B, T2, 80, device=dev)
        tl64, ml64 = torch.full((B,), T1, device=dev), torch.full((B,), T2, device=dev)
        res = {}
        for rep in range(3):
            for name, fa, fe in (("chain", False, False), ("fused", True, True), ("align only", True, False), ("expand only", False, True)):
                m.fuse_align, m.fuse_expand = fa, fe
                with torch.no_grad():
                    t = timeit(lambda: m(text, tl64, mel, ml64), n=20, warm=4)
                res.setdefault(name, []).append(t)
        for k, v in res.items():
            print(f"  forward {prec:7s} {k:12s}: " + " ".join(f"{x:8.1f}" for x in v) + " us")


if __name__ == "__main__":
    t0 = time.time()
    main()
    print(f"({time.time() - t0:.0f} s)")
