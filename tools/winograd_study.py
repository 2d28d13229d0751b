#!/usr/bin/env python3
"""Feasibility study, step 1: would a FLOP-reducing k5 layer -- 1-D Winograd F(2,5) along time, 6 products per
output pair instead of 10 -- keep the path's numerics?  CPU emulation on the oracle (test infrastructure, like
oracle/precision_emulation.py): every k5 Conv1d of the model (5 + 3 + 6 ResConv1d layers, nntts/layers/efts_modules.py:32-35) is
computed as  y = A^T [ (G w) . (B^T x) ]  with the TRANSFORMED operands rounded the way the MFMA would see them (bf16, or hi + lo
bf16 with the lo*lo product dropped) and fp32 accumulation, for two point sets; everything else as in the operand mode.
Prints mel max-abs against the fp32 oracle beside the direct convolution in the same mode.  Zero GPU minutes.

Go criterion of the brief: bf16x3 stays <= 1e-3 AND bf16 within 2x of the direct form's error."""
import os
import sys
from fractions import Fraction

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import efts_oracle as O                      # noqa: E402
from oracle.precision_emulation import split             # noqa: E402


def matrices(points):
    """A^T (2 x 6), G (6 x 5), B^T (6 x 6) of F(2, 5) for five finite points + infinity (Toom-Cook; B^T solved from the defining identity
    sum_j A^T[i,j] G[j,k] B^T[j,t] = [t == i + k], exact in rationals)"""
    m, r = 2, 5
    n = m + r - 1
    p = [Fraction(x) for x in points]
    assert len(p) == n - 1
    AT = [[p[j] ** i for j in range(n - 1)] + [Fraction(int(i == m - 1))] for i in range(m)]
    N = [np.prod([p[j] - p[l] for l in range(n - 1) if l != j]) for j in range(n - 1)]
    G = [[p[j] ** k / N[j] for k in range(r)] for j in range(n - 1)] + [[Fraction(int(k == r - 1)) for k in range(r)]]
    # solve for B^T column by column in float64, then check the identity
    AT_f, G_f = np.array(AT, dtype=np.float64), np.array(G, dtype=np.float64)
    M = np.stack([AT_f[i, :] * G_f[:, k] for i in range(m) for k in range(r)])          # [(i,k), j]
    BT = np.zeros((n, n))
    for t in range(n):
        rhs = np.array([float(t == i + k) for i in range(m) for k in range(r)])
        sol, res, *_ = np.linalg.lstsq(M, rhs, rcond=None)
        assert np.abs(M @ sol - rhs).max() < 1e-9, "not a valid point set"
        BT[:, t] = sol
    return AT_f, G_f, BT


class WinogradMode:
    """patches F.conv1d / F.linear / torch.bmm like precision_emulation.Mode; k5 convolutions take the F(2,5) form"""

    def __init__(self, mode, points=None, scale_rows=True):
        self.mode, self.points = mode, points
        if points is not None:
            AT, G, BT = matrices(points)
            if scale_rows:
                # free diagonal scaling between G and B^T rows (keeps the identity): balance the row norms of B^T to 1-norm 1 per row maximum
                s = np.abs(BT).sum(1)
                BT, G = BT / s[:, None], G * s[:, None]
            self.AT, self.G, self.BT = (torch.tensor(x, dtype=torch.float32) for x in (AT, G, BT))

    def rnd(self, f, a, b):
        if self.mode == "fp32":
            return f(a, b)
        ah, al = split(a)
        bh, bl = split(b)
        if self.mode == "bf16":
            return f(ah, bh)
        return f(ah, bh) + f(ah, bl) + f(al, bh)

    def conv(self, x, w, bias=None, padding=0):
        if self.points is None or w.shape[2] != 5 or padding != 2:
            y = self.rnd(lambda a, b: self.c(a, b, None, padding=padding), x, w)
            return y + (0 if bias is None else bias[None, :, None])
        B, C, T = x.shape
        Te = T + (T & 1)
        xp = F.pad(x, (2, 2 + Te - T))                                  # [B, C, Te + 4]
        tiles = xp.unfold(2, 6, 2)                                      # [B, C, Te/2, 6]
        U = torch.einsum("jt,bcit->jbci", self.BT, tiles)               # transformed input, fp32, then rounded as an MFMA operand
        V = torch.einsum("jk,ock->joc", self.G, w)                      # transformed weights
        Mj = torch.stack([self.rnd(lambda a, b: torch.einsum("oc,bci->boi", a, b), V[j], U[j]) for j in range(6)])   # [6, B, O, Te/2]
        y = torch.einsum("ij,jbot->boti", self.AT, Mj).reshape(B, w.shape[0], Te)[:, :, :T]
        return y + (0 if bias is None else bias[None, :, None])

    def __enter__(self):
        self.c, self.l, self.b = F.conv1d, F.linear, torch.bmm
        F.conv1d = self.conv
        F.linear = lambda x, w, bias=None: self.rnd(lambda a, b: self.l(a, b), x, w) + (0 if bias is None else bias)
        torch.bmm = lambda a, b: self.rnd(self.b, a, b) if self.mode != "bf16" else self._bmm_x3(a, b)
        return self

    def _bmm_x3(self, a, b):                                            # (the product keeps attention on hi/lo operands in both modes)
        ah, al = split(a)
        bh, bl = split(b)
        return self.b(ah, bh) + self.b(ah, bl) + self.b(al, bh)

    def __exit__(self, *a):
        F.conv1d, F.linear, torch.bmm = self.c, self.l, self.b


def main():
    torch.set_num_threads(int(os.environ.get("THREADS", "16")))
    P = O.fill_params()
    here = os.path.dirname(os.path.abspath(__file__))
    g = np.load(os.path.join(here, "..", "tests", "golden", "fwd_full.npz"))
    args = [torch.from_numpy(g[k]) for k in ("text", "text_lengths", "speech", "speech_lengths")]
    sets = {"direct": None, "{0,+-1,+-2,inf}": (0, 1, -1, 2, -2), "{0,+-1,+-1/2,inf}": (0, 1, -1, Fraction(1, 2), Fraction(-1, 2))}
    with torch.no_grad():
        ref = O.forward(P, *args)
        # exactness of the transform itself (fp32 operands): the algebra, and what fp32 transforms cost
        for name, pts in sets.items():
            for mode in ("fp32", "bf16x3", "bf16"):
                if pts is None and mode == "fp32":
                    continue
                for scaled in ((True, False) if pts is not None else (True,)):
                    with WinogradMode(mode, pts, scaled):
                        out = O.forward(P, *args)
                    print(f"{name:20s} {'row-scaled' if scaled else 'plain     '} operands {mode:7s}: " + " ".join(
                        f"{k} {float((out[k] - ref[k]).abs().max()):.2e}" for k in ("mel_pred", "e", "reconst_alpha", "dur_pred")) +
                        f"   (|mel| max {float(ref['mel_pred'].abs().max()):.2f})", flush=True)
    for name, pts in sets.items():
        if pts is not None:
            AT, G, BT = matrices(pts)
            print(f"{name}: max |B^T| row 1-norm {np.abs(BT).sum(1).max():.2f}, max |G| row 1-norm {np.abs(G).sum(1).max():.3f}, "
                  f"max |A^T| row 1-norm {np.abs(AT).sum(1).max():.1f}")


if __name__ == "__main__":
    main()
