"""Lab build -DRC_STAMP of efts_resconv5: per-workgroup start / end stamps (100 MHz constant clock) of back-to-back dependent
launches: dispatch skew, kernel span, and the idle gap between one launch's last workgroup ending and the next one's first
starting.  EFTS_LIB must point at the lab build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
stamp = torch.zeros(64, 1024, dtype=torch.int64, device=dev)
os.environ["EFTS_RC_STAMP"] = hex(stamp.data_ptr())
from efficient_tts_amd import lib as L, ops as P
L.load(); L.require_device()
P.RC_KERNEL = int(os.environ.get("RCK", "0"))     # RCK: 0 / 1 the 8-wave kernel (default), 2 the one-wave-per-SIMD kernel where it applies
C = 512
B, T = int(os.environ.get("PB", 64)), int(os.environ.get("PT", 800))
split = int(os.environ.get("PSPLIT", 1))
rs = P.Rows(B, T)
def plane(sp):
    pl = P.Plane.for_rows(rs, C, sp, dev)
    pl.buf.copy_(torch.randn(pl.buf.numel() // 2, device=dev).mul_(0.5).to(torch.bfloat16).view(torch.uint8).view(pl.buf.shape))
    return pl
xa, xb = plane(split), plane(split)
la, lb = (plane(1), plane(1)) if split == 1 else (None, None)
pw = P.PackedWeight(C, C, 5, split, dev); pw.pack((torch.randn(C, C, 5, device=dev) * 0.02).contiguous())
bias = torch.randn(C, device=dev)
gap = torch.ones(rs.rows, device=dev)
N = 24
with P.stream_scope():
    for rep in range(2):
        for i in range(N):      # a chain: layer i reads what layer i - 1 wrote
            src, dst, sl, dl = (xa, xb, la, lb) if i % 2 == 0 else (xb, xa, lb, la)
            P.resconv5(x=src, x_lo=sl, w=pw, m=rs.rows, n=C, bias=bias, rowmask_ptr=gap.data_ptr(), y=dst, y_lo=dl)
        torch.cuda.synchronize()
st = stamp.cpu().numpy()
groups, _ = P.resconv5_plan(rs.rows, C, 0)
nwg = groups * 2
rows = []
first = (2 * N) % 64 - N           # slot of launch 0 of the second repetition
for i in range(N):
    s = st[(first + i) % 64, :2 * nwg].reshape(nwg, 2)
    rows.append((s[:, 0].min(), s[:, 0].max(), s[:, 1].min(), s[:, 1].max()))
print(f"B={B} T={T} split={split}: {nwg} workgroups; per launch (us): dispatch skew (last start - first start), span (last end - first start), "
      f"end skew (last end - first end), gap to the next launch (its first start - this last end)")
for i in range(N - 1):
    a, b = rows[i], rows[i + 1]
    print(f"  launch {i:2d}: skew {(a[1] - a[0]) / 100:6.2f}  span {(a[3] - a[0]) / 100:7.2f}  end skew {(a[3] - a[2]) / 100:6.2f}  gap {(b[0] - a[3]) / 100:6.2f}")
import numpy as np
s = st[(first + N - 2) % 64, :2 * nwg].reshape(nwg, 2).astype(np.float64)
t0 = s[:, 0].min()
end = (s[:, 1] - t0) / 100
print("one launch, end time of the workgroups (us after the first start): percentiles 0/10/25/50/75/90/100 =",
      " ".join(f"{np.percentile(end, q):.1f}" for q in (0, 10, 25, 50, 75, 90, 100)))
# blockIdx -> (xcd, group, column tile) as the kernel maps it
nb = nwg; q8, r8 = nb >> 3, nb & 7
ids = np.arange(nb); xcd = ids & 7; loc = ids >> 3
v = np.where(xcd < r8, xcd * (q8 + 1), r8 * (q8 + 1) + (xcd - r8) * q8) + loc
grp, col = v // 2, v % 2
for x in range(8):
    m = xcd == x
    print(f"  XCD {x}: {m.sum():3d} workgroups, end mean {end[m].mean():6.1f}  min {end[m].min():6.1f}  max {end[m].max():6.1f}")
for cl in range(2):
    m = (grp % 2) == cl
    print(f"  class {cl}: end mean {end[m].mean():6.1f}  max {end[m].max():6.1f};  column tile {cl}: end mean {end[col == cl].mean():6.1f}")
order = np.argsort(end)
print("  earliest 6 (group, end):", [(int(grp[i]), round(float(end[i]), 1)) for i in order[:6]], " latest 6:", [(int(grp[i]), round(float(end[i]), 1)) for i in order[-6:]])
print("per-XCD mean end (us after the launch's first start) of several launches:")
for li in (2, 8, 14, 20):
    s = st[(first + li) % 64, :2 * nwg].reshape(nwg, 2).astype(np.float64)
    e = (s[:, 1] - s[:, 0].min()) / 100
    print(f"  launch {li:2d}: " + " ".join(f"{e[xcd == x].mean():6.1f}" for x in range(8)))
