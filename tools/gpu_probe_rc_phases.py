"""Lab build -DRC_STAMP=2 of efts_resconv5: shader-clock cycles every wave spends in the four parts of the ping-pong main loop
(read phase up to its barrier, waiting at that barrier, MFMA phase up to its barrier, waiting at that barrier), summed over a
launch, for the first 32 workgroups.  EFTS_LIB must point at the lab build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
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
    with P.stream_scope():
        for i in range(64):
            src, dst, sl, dl = (xa, xb, la, lb) if i % 2 == 0 else (xb, xa, lb, la)
            P.resconv5(x=src, x_lo=sl, w=pw, m=rs.rows, n=C, bias=bias, rowmask_ptr=gap.data_ptr(), y=dst, y_lo=dl)
        torch.cuda.synchronize()
    st = stamp.cpu().numpy()[32:].reshape(32, 32, 8, 4).astype(np.float64).mean(0)      # [wg][wave][4], mean of the last 32 launches
    tot = st.sum(-1)
    print(f"B={B} T={T} split={split}: cycles per launch and wave, mean over 32 workgroups (read / its barrier / MFMA / its barrier / sum)")
    for row, name in ((0, "wave row 0 (waves 0-3)"), (1, "wave row 1 (waves 4-7)")):
        m = st[:, row * 4:(row + 1) * 4].mean((0, 1))
        print(f"  {name}: " + "  ".join(f"{v:9.0f}" for v in m) + f"  {m.sum():9.0f}   = " + " / ".join(f"{100 * v / m.sum():.0f}%" for v in m))
