#!/usr/bin/env python3
"""Golden fixtures for the ctor options outside the shipped YAML (efficient_tts.py:43-48), produced by the REFERENCE itself
(build container only):

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_variants.py [name ...]

use_masking=False (the ctor default), share_text_encoder_key_value=True, use_mel_query_fc=True, delta_e_method_1=False, each on
the ragged (2, 16, 64) case of gen_golden.py with parameter gradients (strided samples + norms); share_text_encoder_key_value and
delta_e_method_1=False also on one free-running utterance.  Writes tests/golden/variant_<name>.npz; only data is stored."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_golden as G  # noqa: E402  (also puts the reference and the oracle on the path)
from gen_golden import EfficientTTSCNN, O  # noqa: E402

VARIANTS = dict(
    nomask=dict(use_masking=False),
    sharekv=dict(share_text_encoder_key_value=True),
    queryfc=dict(use_mel_query_fc=True),
    delta2=dict(delta_e_method_1=False),
    k3=dict(k_size=3),                                              # ResConv1d's kernel size (efts_modules.py:19-46)
    relu=dict(nonlinear_activation="ReLU", nonlinear_activation_params={}),   # ... and its activation
    k7=dict(k_size=7),                                              # kernel sizes past the shipped 5: row spaces with 3 / 5 gap rows,
    k11=dict(k_size=11),                                            # the stacks on efts_gemm's 7- / 11-tap forms
    gelu=dict(nonlinear_activation="GELU", nonlinear_activation_params={}),                 # any other torch.nn activation (efts_modules.py:32-35):
    elu=dict(nonlinear_activation="ELU", nonlinear_activation_params={"alpha": 0.7}),       # a smooth one and one with a parameter
)
# what the oracle's hyper-parameter dict calls an option, where it differs from the reference ctor's keyword
ORACLE_HP = dict(relu=dict(leaky_slope=0.0), gelu=dict(activation=("GELU", {})), elu=dict(activation=("ELU", {"alpha": 0.7})))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    only = set(sys.argv[1:])                                         # (names on the command line: only those fixtures are rewritten)
    for name, opt in VARIANTS.items():
        if only and name not in only:
            continue
        hp = dict(O.DEFAULT_HP, **ORACLE_HP.get(name, opt))
        kw = dict(num_symbols=76, dropout_rate=0.0, use_masking=True, use_weighted_masking=False, sigma=0.01)
        kw.update(opt)
        m = EfficientTTSCNN(**kw)
        P = O.fill_params(hp)
        sd = m.state_dict()
        assert list(sd.keys()) == list(P.keys()), (name, "state_dict key order/names differ")
        m.load_state_dict(P)
        m.eval()
        text, tl, mel, sl = G.make_inputs(11, 2, 16, 64, [16, 11], [64, 50], True)
        if name == "nomask":
            mel = mel + 0.0                                   # zero padding as the collate leaves it
        m.zero_grad()
        ref = G.ref_forward(m, text, tl, mel, sl)
        o = O.forward(P, text, tl, mel, sl, hp)
        for k in ("loss", "mel_loss", "dur_loss", "imv", "e", "reconst_alpha", "mel_pred", "dur_pred", "log_delta_e"):
            print(f"  [{name}] oracle vs reference {k:14s} max-abs {G.maxabs(o[k], ref[k]):.3e}")
        d = dict(text=G.npy(text), text_lengths=G.npy(tl), speech=G.npy(mel), speech_lengths=G.npy(sl))
        for k in ("loss", "mel_loss", "dur_loss", "imv", "e", "dur_pred", "log_delta_e", "mel_pred", "reconst_alpha"):
            d[k] = G.npy(ref[k])
        ref["loss"].backward()
        Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        O.forward(Pg, text, tl, mel, sl, hp)["loss"].backward()
        worst, zero_like = 0.0, []
        gmax_all = max(float(p.grad.abs().max()) for _, p in m.named_parameters())
        for k, p in m.named_parameters():
            g = p.grad
            # a gradient that is identically zero in exact arithmetic (text_encoder_key.bias: softmax over the keys is shift-invariant) is fp noise
            # on both sides: comparing noise with noise relative to its own size says nothing, so such tensors are reported apart
            if float(g.abs().max()) <= 1e-5 * gmax_all:
                zero_like.append((k, float(g.abs().max()), float(Pg[k].grad.abs().max())))
            else:
                worst = max(worst, G.maxabs(g, Pg[k].grad) / float(g.abs().max()))
            flat = G.npy(g).reshape(-1)
            if flat.size > 1024:
                step = flat.size // 512
                d["grad_stride:" + k] = np.int64(step)
                flat = flat[::step]
            d["grad:" + k] = flat
            d["gradnorm:" + k] = np.float64(g.double().norm())
        print(f"  [{name}] oracle vs reference param-grad worst rel-to-max {worst:.3e}"
              + "".join(f"; {k}: zero in exact arithmetic (|grad| max: reference {a:.1e}, oracle {b:.1e}), not in the figure" for k, a, b in zero_like))
        if name in ("sharekv", "delta2", "k3", "k7", "k11", "gelu", "elu"):                # free-running path: value = key (:252-253) / positions from 0 (:261-265) / k3 stacks
            ids = torch.randint(1, 76, (1, 23), generator=torch.Generator().manual_seed(5))
            m.remove_weight_norm()
            with torch.no_grad():
                mel_pred, ralpha = m.inference(ids)
                oi = O.inference(P, ids, hp)
            print(f"  [{name}] inference T2={mel_pred.shape[1]} oracle-vs-ref mel {G.maxabs(oi['mel_pred'], mel_pred):.3e}")
            d["inf_text"] = G.npy(ids); d["inf_mel_pred"] = G.npy(mel_pred); d["inf_t2"] = np.int64(mel_pred.shape[1])
        np.savez_compressed(os.path.join(G.OUT, f"variant_{name}.npz"), **d)
        print(f"wrote variant_{name}.npz loss={float(ref['loss']):.6f}")


if __name__ == "__main__":
    main()
