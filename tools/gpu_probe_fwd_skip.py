"""upper bounds for the teacher-forced forward (B=64, 128 x 800): time per call with groups of launches turned into no-ops
(results are garbage then: what REMOVING them could return at most; the skipped kernels' outputs stay what an earlier full pass left)
python tools/gpu_probe_fwd_skip.py [bf16|bf16x3]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficient_tts_amd import EfficientTTSCNN, lib as L, ops as O

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda:0")
B, T1, T2 = 64, 128, 800
m = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, sigma=0.01, precision=prec).to(dev).eval()
m.graphs = False
text = torch.randint(0, 76, (B, T1), device=dev)
mel = torch.randn(B, T2, 80, device=dev)
tl, ml = torch.full((B,), T1, device=dev), torch.full((B,), T2, device=dev)
lib = L.load()
real_gemm = O.gemm
rows1, rows2 = O.Rows(B, T1).rows, O.Rows(B, T2).rows
GROUPS = {
    "none": ([], None),
    "q.k^T (gemm 1 x 800 x 128)": ([], lambda kw: kw.get("taps", 1) == 1 and kw["n"] == T1 and kw["m"] == T2),
    "key + value projections": ([], lambda kw: kw.get("taps", 1) == 1 and kw["m"] == rows1 and kw["n"] == 512),
    "duration predictor convs (gemm k3)": ([], lambda kw: kw.get("taps", 1) == 3),
    "duration predictor LayerNorms": (["efts_layernorm_rows", "efts_layernorm_dot"], None),
    "whole duration predictor": (["efts_layernorm_rows", "efts_layernorm_dot"], lambda kw: kw.get("taps", 1) == 3),
    "mel head (gemm 512 -> 80)": ([], lambda kw: kw.get("taps", 1) == 1 and kw["m"] == rows2 and kw["n"] == 80),
    "prenet (efts_frame_linear)": (["efts_frame_linear"], None),
    "efts_expand": (["efts_expand"], None),
    "efts_imv_align": (["efts_imv_align"], None),
    "embed_conv + masks": (["efts_embed_conv", "efts_row_masks_pair"], None),
    "losses": (["efts_masked_losses"], None),
    "text riders + text layer 1 (all efts_resconv5 at text length)": ("text", None),
}
saved = {n: getattr(lib, n) for n in ("efts_layernorm_rows", "efts_layernorm_dot", "efts_frame_linear", "efts_expand", "efts_imv_align", "efts_embed_conv",
                                      "efts_row_masks_pair", "efts_masked_losses", "efts_resconv5", "efts_resconv5_multi")}


def timeit(n=30, warm=6):
    with torch.no_grad():
        for _ in range(warm):
            m(text, tl, mel, ml)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            m(text, tl, mel, ml)
        e.record()
        torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


with torch.no_grad():
    for _ in range(3):
        m(text, tl, mel, ml)
for rep in range(2):
    for name, (entries, pred) in GROUPS.items():
        for k, v in saved.items():
            setattr(lib, k, v)
        O.gemm = real_gemm
        if entries == "text":
            def multi(arr, n, st, _f=saved["efts_resconv5_multi"]):
                return _f(arr, 1, st)                      # the mel-encoder layer alone
            def single(g, st, _f=saved["efts_resconv5"]):
                return 0 if g._obj.m < 20000 else _f(g, st)
            lib.efts_resconv5_multi, lib.efts_resconv5 = multi, single
        else:
            for n_ in entries:
                setattr(lib, n_, lambda *a: 0)
        if pred is not None:
            O.gemm = lambda _p=pred, **kw: None if _p(kw) else real_gemm(**kw)
        print(f"BOUND {prec} {name:60s}: {timeit():8.1f} us", flush=True)
