"""efts_act_bwd at mel length (B=32 x 800 rows x 512): time by mode and with / without the bias-gradient atomics."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from efficient_tts_amd import lib as L, ops as P
dev = torch.device("cuda:0"); L.load(); L.require_device()
B, T, C = int(os.environ.get("PB", 32)), 800, 512
rs = P.Rows(B, T)
G = torch.randn(rs.rows, C, device=dev); y = torch.randn(rs.rows, C, device=dev); x = torch.randn(rs.rows, C, device=dev)
sg = torch.randint(0, 255, (rs.rows, C // 8), dtype=torch.uint8, device=dev)
gap = torch.ones(rs.rows, device=dev)
pl = P.Plane.for_rows(rs, C, 1, dev)
db = torch.zeros(C, device=dev)
def t(fn, n=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
with P.stream_scope():
    lib = L.load()
    for mode, yy, xx in ((4, sg, None), (1, y, x), (0, None, None)):
        for bias in (True, False):
            fn = lambda: L.check(lib.efts_act_bwd(G.data_ptr(), None if yy is None else yy.data_ptr(), None if xx is None else xx.data_ptr(), gap.data_ptr(),
                                                  0.1, mode, None, pl.ptr, pl.ld, 1, db.data_ptr() if bias else None, rs.rows, C, P._stream()), "act_bwd")
            print(f"mode {mode} dbias {bias}: {t(fn):.1f} us", flush=True)
