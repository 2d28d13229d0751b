"""Lab build -DRC_STAMP=3: per workgroup, kernel start / end of tile 0's main loop / end of tile 0's epilogue / kernel end (us)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
dev = torch.device("cuda:0")
stamp = torch.zeros(64, 1024, dtype=torch.int64, device=dev)
os.environ["EFTS_RC_STAMP"] = hex(stamp.data_ptr())
from efficient_tts_amd import lib as L, ops as P
L.load(); L.require_device()
P.RC_KERNEL = int(os.environ.get("RCK", "0"))     # RCK: 0 / 1 the 8-wave kernel (default), 2 the one-wave-per-SIMD kernel where it applies
C = 512
B, T = int(os.environ.get("PB", 64)), int(os.environ.get("PT", 800))
for split in (1, 2):
    rs = P.Rows(B, T)
    def plane(sp):
        pl = P.Plane.for_rows(rs, C, sp, dev)
        pl.buf.copy_(torch.randn(pl.buf.numel() // 2, device=dev).mul_(0.5).to(torch.bfloat16).view(torch.uint8).view(pl.buf.shape))
        return pl
    xa, xb = plane(split), plane(split)
    la, lb = (plane(1), plane(1)) if split == 1 else (None, None)
    pw = P.PackedWeight(C, C, 5, split, dev); pw.pack((torch.randn(C, C, 5, device=dev) * 0.02).contiguous())
    bias = torch.randn(C, device=dev); gap = torch.ones(rs.rows, device=dev)
    stamp.zero_()
    with P.stream_scope():
        for i in range(64):
            src, dst, sl, dl = (xa, xb, la, lb) if i % 2 == 0 else (xb, xa, lb, la)
            P.resconv5(x=src, x_lo=sl, w=pw, m=rs.rows, n=C, bias=bias, rowmask_ptr=gap.data_ptr(), y=dst, y_lo=dl)
        torch.cuda.synchronize()
    groups, classes = P.resconv5_plan(rs.rows, C, 0)
    nwg = groups * 2
    st = stamp.cpu().numpy()[32:, :4 * nwg].reshape(32, nwg, 4).astype(np.float64) / 100.0     # us
    t0 = st[:, :, 0].min(axis=1, keepdims=True)
    print(f"B={B} T={T} split={split} plan {classes}: mean over {nwg} workgroups x 32 launches (us)")
    print(f"  start after the launch's first start {np.mean(st[:, :, 0] - t0):6.1f}   tile 0 main loop {np.mean(st[:, :, 1] - st[:, :, 0]):6.1f}   "
          f"tile 0 epilogue {np.mean(st[:, :, 2] - st[:, :, 1]):6.1f}   rest (tile 1 or drain) {np.mean(st[:, :, 3] - st[:, :, 2]):6.1f}   "
          f"whole workgroup {np.mean(st[:, :, 3] - st[:, :, 0]):6.1f}   launch span {np.mean(st[:, :, 3].max(axis=1) - t0[:, 0]):6.1f}")
