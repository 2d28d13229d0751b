"""Emulation of the MFMA operand modes on the CPU oracle  --  TEST INFRASTRUCTURE.

Patches torch's conv1d / linear / bmm inside a `with Mode(conv_mode, attn_mode):` block so that the
oracle (oracle/efts_oracle.py) computes with
  "bf16"   : operands rounded to bf16, fp32 accumulate
  "bf16x3" : a = a_hi + a_lo (two bf16), product = hi*hi + hi*lo + lo*hi, fp32 accumulate
  "fp32"   : unchanged
Used by tools/attic/precision_study.py (design study) and by the gradient parity tests: the HIP backward
must agree tightly with autograd of the oracle run in the SAME operand mode, while the distance to
the fp32 oracle is the (documented) sensitivity of the alignment block to that mode.
"""
import torch
import torch.nn.functional as F


def split(x):
    hi = x.to(torch.bfloat16).float()
    lo = (x - hi).to(torch.bfloat16).float()
    return hi, lo

class Mode:
    def __init__(self, conv_mode, attn_mode):
        self.conv_mode, self.attn_mode = conv_mode, attn_mode
    def op(self, f, a, b, mode):
        if mode == "fp32":
            return f(a, b)
        ah, al = split(a); bh, bl = split(b)
        if mode == "bf16":
            return f(ah, bh)
        return f(ah, bh) + f(ah, bl) + f(al, bh)
    def __enter__(self):
        self.c, self.l, self.b = F.conv1d, F.linear, torch.bmm
        F.conv1d = lambda x, w, bias=None, padding=0: self.op(lambda a, b: self.c(a, b, None, padding=padding), x, w, self.conv_mode) + (0 if bias is None else bias[None, :, None])
        F.linear = lambda x, w, bias=None: self.op(lambda a, b: self.l(a, b), x, w, self.conv_mode) + (0 if bias is None else bias)
        torch.bmm = lambda a, b: self.op(self.b, a, b, self.attn_mode)
    def __exit__(self, *a):
        F.conv1d, F.linear, torch.bmm = self.c, self.l, self.b

