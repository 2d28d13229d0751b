"""CPU: the oracle (oracle/efts_oracle.py) against fixtures produced by the reference itself
(tools/gen_golden.py imported nntts.models.EfficientTTSCNN in the build container)."""
import os

import numpy as np
import pytest
import torch

from oracle import efts_oracle as O

# fp32 re-association noise between two CPU formulations of the same maths (measured when the
# fixtures were generated: mel_pred 1.8e-4, e 4.9e-4 at T2=1200); the product gate is 1e-3.
TOL = dict(loss=2e-5, mel_loss=2e-5, dur_loss=2e-5, imv=3e-4, e=1e-3, dur_pred=2e-5,
           log_delta_e=3e-4, mel_pred=5e-4, reconst_alpha=1e-4)


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


@pytest.fixture(scope="module")
def params():
    return O.fill_params()


@pytest.mark.parametrize("case", ["fwd_tiny", "fwd_small", "fwd_full", "fwd_long"])
def test_forward_matches_reference(golden_dir, params, case):
    g = _load(golden_dir, case)
    out = O.forward(params, torch.from_numpy(g["text"]), torch.from_numpy(g["text_lengths"]),
                    torch.from_numpy(g["speech"]), torch.from_numpy(g["speech_lengths"]))
    st, sa = int(g["mel_pred_stride"]), int(g["alpha_stride"])
    got = dict(out)
    got["mel_pred"] = out["mel_pred"][:, ::st, :]
    got["reconst_alpha"] = out["reconst_alpha"][:, ::sa, ::sa]
    for k, tol in TOL.items():
        ref = torch.from_numpy(np.asarray(g[k]))
        scale = max(1.0, float(ref.abs().max())) if k in ("loss", "mel_loss", "dur_loss") else 1.0
        err = float((got[k].detach() - ref).abs().max())
        assert err <= tol * scale, f"{case}:{k} max-abs {err:.3e} > {tol * scale:.1e}"
    assert abs(float(out["mel_pred"].double().sum()) - float(g["mel_pred_sum"])) <= 1e-4 * float(g["mel_pred_abssum"])


def test_param_grads_match_reference(golden_dir, params):
    g = _load(golden_dir, "fwd_tiny")
    P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    O.forward(P, torch.from_numpy(g["text"]), torch.from_numpy(g["text_lengths"]),
              torch.from_numpy(g["speech"]), torch.from_numpy(g["speech_lengths"]))["loss"].backward()
    for k, p in P.items():
        ref = g["grad:" + k]
        flat = p.grad.reshape(-1).numpy()
        if "grad_stride:" + k in g.files:
            flat = flat[:: int(g["grad_stride:" + k])]
        # text_encoder_key.bias has an identically-zero gradient (softmax shift invariance)
        tol = 2e-5 * max(float(np.abs(ref).max()), 1e-2)
        assert np.abs(flat - ref).max() <= tol, k
        assert abs(float(p.grad.double().norm()) - float(g["gradnorm:" + k])) <= 1e-4 * float(g["gradnorm:" + k]) + 1e-6


def test_inference_matches_reference(golden_dir, params):
    g = _load(golden_dir, "inference_lj")
    for n in range(4):
        out = O.inference(params, torch.from_numpy(g[f"text{n}"]))
        assert out["t2"] == int(g[f"t2_{n}"])
        assert float((out["mel_pred"][:, ::2, :] - torch.from_numpy(g[f"mel_pred{n}"])).abs().max()) <= 1e-4
        assert float((out["reconst_alpha"][:, ::4, ::4] - torch.from_numpy(g[f"reconst_alpha{n}"])).abs().max()) <= 1e-5


def test_weight_norm_fold_is_identity_on_outputs(params):
    """remove_weight_norm (efficient_tts.py:400-409) must not change outputs."""
    folded = {}
    for k, v in params.items():
        if k.endswith("weight_g"):
            continue
        if k.endswith("weight_v"):
            folded[k[:-2]] = O.weight_norm_fold(v, params[k[:-1] + "g"])
        else:
            folded[k] = v
    ids = torch.randint(0, 76, (1, 23), generator=torch.Generator().manual_seed(3))
    a, b = O.inference(params, ids), O.inference(folded, ids)
    assert a["t2"] == b["t2"] and torch.equal(a["mel_pred"], b["mel_pred"])


def test_train3_matches_reference(golden_dir, params):
    """3 x (fwd, bwd, clip 1.0, Adam-amsgrad, WarmupLR) of trainer.py:139-160 with the oracle's
    own optimizer restatement."""
    g = _load(golden_dir, "train3")
    t = _load(golden_dir, "fwd_tiny")
    text, tl = torch.from_numpy(t["text"]), torch.from_numpy(t["text_lengths"])
    mel, sl = torch.from_numpy(t["speech"]), torch.from_numpy(t["speech_lengths"])
    P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    st = {k: [torch.zeros_like(v), torch.zeros_like(v), torch.zeros_like(v)] for k, v in P.items()}
    for step in range(1, 4):
        for p in P.values():
            p.grad = None
        loss = O.forward(P, text, tl, mel, sl)["loss"]
        loss.backward()
        gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in P.values()))
        assert abs(float(loss) - g["losses"][step - 1]) <= 2e-4 * g["losses"][step - 1]
        assert abs(float(gn) - g["gnorm"][step - 1]) <= 1e-3 * g["gnorm"][step - 1]
        coef = min(1.0, 1.0 / (float(gn) + 1e-6))
        lr = O.warmup_lr(1e-3, step, 4000)
        assert abs(lr - g["lrs"][step - 1]) < 1e-12
        with torch.no_grad():
            for k, p in P.items():
                O.adam_amsgrad_step(p, p.grad * coef, *st[k], step=step, lr=lr)
    for k, p in P.items():
        ref = g["param:" + k]
        flat = p.detach().reshape(-1).numpy()[:: int(g["param_stride:" + k])]
        assert np.abs(flat - ref).max() <= 2e-6 + 1e-4 * np.abs(ref).max(), k


VARIANTS = dict(nomask=dict(use_masking=False), sharekv=dict(share_text_encoder_key_value=True), queryfc=dict(use_mel_query_fc=True),
                delta2=dict(delta_e_method_1=False), k3=dict(k_size=3), relu=dict(leaky_slope=0.0), k7=dict(k_size=7), k11=dict(k_size=11),
                gelu=dict(activation=("GELU", {})), elu=dict(activation=("ELU", {"alpha": 0.7})))     # (relu: the reference ctor's nonlinear_activation="ReLU")


@pytest.mark.parametrize("name", list(VARIANTS))
def test_ctor_option_variants_match_reference(golden_dir, name):
    """the oracle's restatement of the ctor options outside the shipped YAML (efficient_tts.py:43-48) against fixtures the
    reference produced with each of them (tools/gen_golden_variants.py): outputs, losses, parameter gradients, key set"""
    g = _load(golden_dir, "variant_" + name)
    hp = dict(O.DEFAULT_HP, **VARIANTS[name])
    P = {k: v.clone().requires_grad_(True) for k, v in O.fill_params(hp).items()}
    assert {k for k in g.files if k.startswith("grad:")} == {"grad:" + k for k in P}
    out = O.forward(P, torch.from_numpy(g["text"]), torch.from_numpy(g["text_lengths"]), torch.from_numpy(g["speech"]),
                    torch.from_numpy(g["speech_lengths"]), hp)
    for k, tol in TOL.items():
        ref = torch.from_numpy(np.asarray(g[k]))
        scale = max(1.0, float(ref.abs().max())) if k in ("loss", "mel_loss", "dur_loss") else 1.0
        assert float((out[k].detach() - ref).abs().max()) <= tol * scale, (name, k)
    out["loss"].backward()
    for k, p in P.items():
        ref = g["grad:" + k]
        flat = p.grad.reshape(-1).numpy()
        if "grad_stride:" + k in g.files:
            flat = flat[:: int(g["grad_stride:" + k])]
        noise = 1e-7 * max(100.0, float(g["loss"]))                       # (fp noise grows with the loss: 863 with k_size 11, ~50-130 otherwise)
        if k == "text_encoder_key.bias" and np.abs(ref).max() <= noise:   # identically zero (softmax shift invariance) unless the value
            assert np.abs(flat).max() <= noise                            # shares the key projection: fp noise on both sides
            continue
        assert np.abs(flat - ref).max() <= 2e-4 * max(float(np.abs(ref).max()), 1e-3), (name, k)
    if name in ("sharekv", "delta2", "k3", "k7", "k11", "gelu", "elu"):
        o = O.inference(O.fill_params(hp), torch.from_numpy(g["inf_text"]), hp)
        assert o["mel_pred"].shape[1] == int(g["inf_t2"])
        assert float((o["mel_pred"] - torch.from_numpy(g["inf_mel_pred"])).abs().max()) <= 5e-4
