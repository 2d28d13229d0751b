#!/usr/bin/env python3
"""Generate golden fixtures by running the REFERENCE itself (build container only).

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py

Imports ``nntts.models.EfficientTTSCNN`` from /root/reference, overwrites every
parameter with the name-keyed deterministic fill of ``oracle/efts_oracle.py``,
runs forward / backward / inference / 3 Adam steps on seeded inputs and stores
inputs + expected outputs as small ``.npz`` files under ``tests/golden/``.
Only data is stored; no reference source travels.  It also cross-checks the
oracle against the reference and prints the max-abs differences.
"""
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
warnings.filterwarnings("ignore")

from nntts.models import EfficientTTSCNN  # noqa: E402  (the reference)
from oracle import efts_oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
FILELIST = "/root/reference/egs/lj/data/nv_taco2_filelists/ljs_audio_phnseq_test_filelist.txt"
PHNSET = "/root/reference/egs/lj/data/nv_taco2_filelists/g2p_en_phnset.txt"


def build_reference():
    m = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True,
                        use_weighted_masking=False, sigma=0.01)
    P = O.fill_params()
    sd = m.state_dict()
    assert list(sd.keys()) == list(P.keys()), "state_dict key order/names differ"
    for k in sd:
        assert tuple(sd[k].shape) == tuple(P[k].shape), k
    m.load_state_dict(P)
    m.eval()
    return m, P


def make_inputs(seed, B, T1, T2, tl, sl, logmel):
    g = torch.Generator().manual_seed(seed)
    text = torch.randint(0, 76, (B, T1), generator=g)
    mel = torch.randn(B, T2, 80, generator=g)
    if logmel:
        mel = (-4.0 + 2.0 * mel).clamp(-11.5, 2.0)
    tl = torch.tensor(tl, dtype=torch.int64)
    sl = torch.tensor(sl, dtype=torch.int64)
    for b in range(B):                      # collate zero-pads (taco2_data.py:101-139)
        text[b, tl[b]:] = 0
        mel[b, sl[b]:] = 0
    return text, tl, mel, sl


def ref_forward(m, text, tl, mel, sl):
    """Run the reference forward, capturing e / dur_pred / log_delta_e via hooks."""
    cap = {}
    orig_gap = m.get_aligned_positions

    def gap(*a, **k):
        r = orig_gap(*a, **k)
        cap["e"] = r.squeeze(-1)
        return r
    m.get_aligned_positions = gap
    h = m.duration_predictor.register_forward_hook(lambda mod, i, o: cap.__setitem__("dur_pred", o))
    orig_crit = m.criterion.forward

    def crit(after, before, d_outs, ys, ds, il, ol):
        cap["log_delta_e"] = ds
        r = orig_crit(after, before, d_outs, ys, ds, il, ol)
        cap["mel_loss"], cap["dur_loss"] = r
        return r
    m.criterion.forward = crit
    loss, stats, imv, ralpha, mel_pred, _ = m(text, tl, mel, sl)
    h.remove()
    m.criterion.forward = orig_crit
    m.get_aligned_positions = orig_gap
    return dict(loss=loss, mel_loss=cap["mel_loss"], dur_loss=cap["dur_loss"], imv=imv, e=cap["e"],
                reconst_alpha=ralpha, mel_pred=mel_pred, dur_pred=cap["dur_pred"],
                log_delta_e=cap["log_delta_e"])


def npy(t):
    return t.detach().cpu().numpy()


def maxabs(a, b):
    return float((a.detach() - b.detach()).abs().max())


def check_oracle(tag, ref, P, text, tl, mel, sl):
    o = O.forward(P, text, tl, mel, sl)
    for k in ("loss", "mel_loss", "dur_loss", "imv", "e", "reconst_alpha", "mel_pred", "dur_pred", "log_delta_e"):
        print(f"  [{tag}] oracle vs reference {k:14s} max-abs {maxabs(o[k], ref[k]):.3e}")
    return o


def case_forward(m, P, name, seed, B, T1, T2, tl, sl, logmel, with_grads, sub):
    text, tl, mel, sl = make_inputs(seed, B, T1, T2, tl, sl, logmel)
    m.zero_grad()
    ref = ref_forward(m, text, tl, mel, sl)
    check_oracle(name, ref, P, text, tl, mel, sl)
    d = dict(text=npy(text), text_lengths=npy(tl), speech=npy(mel), speech_lengths=npy(sl),
             loss=npy(ref["loss"]), mel_loss=npy(ref["mel_loss"]), dur_loss=npy(ref["dur_loss"]),
             imv=npy(ref["imv"]), e=npy(ref["e"]), dur_pred=npy(ref["dur_pred"]),
             log_delta_e=npy(ref["log_delta_e"]))
    st, sa = sub
    d["mel_pred_stride"] = np.int64(st)
    d["alpha_stride"] = np.int64(sa)
    d["mel_pred"] = npy(ref["mel_pred"])[:, ::st, :]
    d["reconst_alpha"] = npy(ref["reconst_alpha"])[:, ::sa, ::sa]
    d["mel_pred_sum"] = np.float64(ref["mel_pred"].double().sum())
    d["mel_pred_abssum"] = np.float64(ref["mel_pred"].double().abs().sum())
    d["reconst_alpha_sum"] = np.float64(ref["reconst_alpha"].double().sum())
    if with_grads:
        ref["loss"].backward()
        # oracle grads through autograd of the restatement
        Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        O.forward(Pg, text, tl, mel, sl)["loss"].backward()
        worst, zero_like = 0.0, []
        gmax_all = max(float(p.grad.abs().max()) for _, p in m.named_parameters())
        for k, p in m.named_parameters():
            g = p.grad
            # a gradient that is identically zero in exact arithmetic (text_encoder_key.bias: softmax over the keys is shift-invariant) is fp noise
            # on both sides: comparing noise with noise relative to its own size says nothing, so such tensors are reported apart
            if float(g.abs().max()) <= 1e-5 * gmax_all:
                zero_like.append((k, float(g.abs().max()), float(Pg[k].grad.abs().max())))
            else:
                worst = max(worst, maxabs(g, Pg[k].grad) / float(g.abs().max()))
            flat = npy(g).reshape(-1)
            if flat.size > 4096:                       # strided samples for the big conv tensors
                step = flat.size // 2048
                d["grad_stride:" + k] = np.int64(step)
                flat = flat[::step]
            d["grad:" + k] = flat
            d["gradnorm:" + k] = np.float64(g.double().norm())
        print(f"  [{name}] oracle vs reference param-grad worst rel-to-max {worst:.3e}"
              + "".join(f"; {k}: zero in exact arithmetic (|grad| max: reference {a:.1e}, oracle {b:.1e}), not in the figure" for k, a, b in zero_like))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(f"wrote {name}.npz  loss={float(ref['loss']):.6f}")


def case_inference(m, P):
    phn = {s.strip(): i for i, s in enumerate(open(PHNSET))}
    lines = [l.strip().split("|")[1] for l in open(FILELIST)][:10]   # inference.py:97 takes the first 10
    d = {}
    m.remove_weight_norm()
    for n, line in enumerate(lines[:4]):
        ids = torch.tensor([[phn[p] for p in line.split()]], dtype=torch.int64)
        with torch.no_grad():
            mel_pred, ralpha = m.inference(ids)
            o = O.inference(P, ids)
        print(f"  [inference {n}] T1={ids.shape[1]} T2={mel_pred.shape[1]} oracle-vs-ref mel {maxabs(o['mel_pred'], mel_pred):.3e}"
              f" alpha {maxabs(o['reconst_alpha'], ralpha):.3e}")
        d[f"text{n}"] = npy(ids)
        d[f"t2_{n}"] = np.int64(mel_pred.shape[1])
        d[f"mel_pred{n}"] = npy(mel_pred)[:, ::2, :]
        d[f"mel_pred_sum{n}"] = np.float64(mel_pred.double().sum())
        d[f"reconst_alpha{n}"] = npy(ralpha)[:, ::4, ::4]
    for n, line in enumerate(lines):                  # ids only, for the config-1 CPU plumbing bench
        d[f"ids{n}"] = np.array([phn[p] for p in line.split()], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "inference_lj.npz"), **d)
    print("wrote inference_lj.npz")


def case_train(P):
    """3 steps of trainer.py:139-160 on the tiny batch with the YAML optimizer (:34-44)."""
    m = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True,
                        use_weighted_masking=False, sigma=0.01)
    m.load_state_dict(P)
    m.eval()                                          # DurationPredictor's Dropout(0.1) off
    text, tl, mel, sl = make_inputs(11, 2, 16, 64, [16, 11], [64, 50], True)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-9, weight_decay=1e-5, amsgrad=True)
    warm = 4000
    d = dict(losses=[], gnorm=[], lrs=[])
    for step in range(1, 4):
        lr = O.warmup_lr(1e-3, step, warm)            # WarmupLR: lr used at step s has step_num = s
        for grp in opt.param_groups:
            grp["lr"] = lr
        loss, stats, *_ = m(text=text, text_lengths=tl, speech=mel, speech_lengths=sl)
        opt.zero_grad()
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        d["losses"].append(float(loss)); d["gnorm"].append(float(gn)); d["lrs"].append(lr)
    out = {k: np.array(v, dtype=np.float64) for k, v in d.items()}
    for k, p in m.named_parameters():
        flat = npy(p).reshape(-1)
        step = max(1, flat.size // 512)
        out["param_stride:" + k] = np.int64(step)
        out["param:" + k] = flat[::step]
    np.savez_compressed(os.path.join(OUT, "train3.npz"), **out)
    print("wrote train3.npz losses", d["losses"], "gnorm", d["gnorm"])


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    m, P = build_reference()
    case_forward(m, P, "fwd_tiny", 11, 2, 16, 64, [16, 11], [64, 50], True, True, (1, 1))
    case_forward(m, P, "fwd_small", 12, 3, 40, 200, [40, 33, 21], [200, 170, 97], False, False, (2, 2))
    case_forward(m, P, "fwd_full", 13, 2, 128, 800, [128, 100], [800, 650], True, False, (8, 8))
    case_forward(m, P, "fwd_long", 14, 1, 128, 1200, [128], [1200], False, False, (8, 8))
    case_train(P)
    case_inference(m, P)


if __name__ == "__main__":
    main()
