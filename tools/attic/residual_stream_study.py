#!/usr/bin/env python3
"""CPU emulation (oracle): what does the storage format of the residual stream between the ResConv1d layers cost?
  fp32  : the stream stays fp32 (round 1)
  hilo  : the stream is rounded to hi + lo (two bf16, 16 mantissa bits) after every layer -- what efts_resconv5 stores
  bf16  : the stream is rounded to bf16 after every layer (plain bf16 activations)
for the bf16 and bf16x3 operand modes, only the mel-length stacks (mel encoder, decoder) or all three.
Prints max-abs error of mel_pred vs the fp32 oracle on the golden inputs."""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import efts_oracle as O
from oracle.precision_emulation import Mode, split

STREAM = {"fmt": "fp32", "blocks": ("mel_encoder", "decoder")}

def rnd(x):
    if STREAM["fmt"] == "fp32":
        return x
    hi, lo = split(x)
    return hi if STREAM["fmt"] == "bf16" else hi + lo

def res_conv_block(x, P, blk, n_layers, slope):
    on = blk in STREAM["blocks"]
    for i in range(n_layers):
        p = f"{blk}.layers.{i}.conv.0."
        w = O.conv_weight(P, p)
        x = x + F.leaky_relu(F.conv1d(x, w, P[p + "bias"], padding=(w.shape[-1] - 1) // 2), slope)
        if on:
            x = rnd(x)
    return x

O.res_conv_block = res_conv_block

def main():
    P = O.fill_params()
    for case in ("fwd_small", "fwd_full"):
        g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", case + ".npz"))
        args = [torch.from_numpy(g[k]) for k in ("text", "text_lengths", "speech", "speech_lengths")]
        with torch.no_grad():
            STREAM["fmt"] = "fp32"
            ref = O.forward(P, *args)
            for cm in ("bf16x3", "bf16"):
                for fmt in ("fp32", "hilo", "bf16"):
                    for blocks in (("mel_encoder", "decoder"), ("text_encoder", "mel_encoder", "decoder")):
                        STREAM["fmt"], STREAM["blocks"] = fmt, blocks
                        with Mode(cm, "bf16x3"):
                            out = O.forward(P, *args)
                        print(case, f"conv={cm:7s} stream={fmt:5s} stacks={len(blocks)}", " ".join(
                            f"{k}={float((out[k]-ref[k]).abs().max()):.2e}" for k in ("mel_pred", "imv", "reconst_alpha", "loss")), flush=True)
if __name__ == "__main__":
    main()
