#!/usr/bin/env python3
"""CPU emulation of the two MFMA operand modes on the oracle (design study for DESIGN.md):
  bf16   : operands rounded to bf16, fp32 accumulate
  bf16x3 : a = a_hi + a_lo (two bf16), product = hi*hi + hi*lo + lo*hi, fp32 accumulate
Prints max-abs error of each output vs the fp32 oracle on the golden inputs."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import efts_oracle as O

from oracle.precision_emulation import Mode  # noqa: E402


def main():
    P = O.fill_params()
    for case in ("fwd_tiny", "fwd_full"):
        g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", case + ".npz"))
        args = [torch.from_numpy(g[k]) for k in ("text", "text_lengths", "speech", "speech_lengths")]
        with torch.no_grad():
            ref = O.forward(P, *args)
            for cm, am in (("bf16x3", "bf16x3"), ("bf16", "bf16x3"), ("bf16", "bf16")):
                with Mode(cm, am):
                    out = O.forward(P, *args)
                print(case, f"conv={cm:7s} attn={am:7s}", " ".join(
                    f"{k}={float((out[k]-ref[k]).abs().max()):.2e}" for k in ("mel_pred", "imv", "e", "reconst_alpha", "dur_pred", "loss")),
                    f"|mel|max={float(ref['mel_pred'].abs().max()):.2f}")
if __name__ == "__main__":
    main()
