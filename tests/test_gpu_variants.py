"""GPU (-m gpu): the ctor options outside the shipped YAML -- use_masking=False (the reference's ctor default,
efficient_tts.py:43), share_text_encoder_key_value=True (:72-75, :150-153, :252-253), use_mel_query_fc=True (:90-95, :163-164),
delta_e_method_1=False (:205-213, :261-265) --
against fixtures the REFERENCE produced for each of them (tools/gen_golden_variants.py): forward outputs, losses, parameter
gradients of the fused training pass, state_dict layout, and the free-running path where the option touches it."""
import os

import numpy as np
import pytest
import torch

from oracle import efts_oracle as O

pytestmark = pytest.mark.gpu
VARIANTS = dict(nomask=dict(use_masking=False), sharekv=dict(share_text_encoder_key_value=True), queryfc=dict(use_mel_query_fc=True),
                delta2=dict(delta_e_method_1=False), k3=dict(k_size=3), relu=dict(nonlinear_activation="ReLU", nonlinear_activation_params={}),
                gelu=dict(nonlinear_activation="GELU", nonlinear_activation_params={}),
                elu=dict(nonlinear_activation="ELU", nonlinear_activation_params={"alpha": 0.7}),  # (activations outside the contraction epilogues: csrc/efts_act.hip)
                k7=dict(k_size=7), k11=dict(k_size=11))            # (kernel sizes past the shipped 5: wider gaps in the row spaces, efts_gemm's 7- / 11-tap forms)
ORACLE_HP = dict(relu=dict(leaky_slope=0.0), gelu=dict(activation=("GELU", {})), elu=dict(activation=("ELU", {"alpha": 0.7})))           # the oracle's name for an option where it differs from the ctor keyword
MEL_TOL = 1e-3


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _model(opt):
    from efficient_tts_amd import EfficientTTSCNN
    kw = dict(num_symbols=76, dropout_rate=0.0, use_masking=True, sigma=0.01, precision="bf16x3")
    kw.update(opt)
    m = EfficientTTSCNN(**kw)
    name = next(k for k, v in VARIANTS.items() if v == opt)
    P = O.fill_params(dict(O.DEFAULT_HP, **ORACLE_HP.get(name, opt)))
    assert list(m.state_dict().keys()) == list(P.keys())          # the reference's key set and order for this option
    m.load_state_dict(P)
    return m.to(_dev()).eval()


@pytest.mark.parametrize("name", list(VARIANTS))
def test_variant_forward_matches_reference_golden(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"variant_{name}.npz"))
    m = _model(VARIANTS[name])
    args = [torch.from_numpy(g[k]).to(_dev()) for k in ("text", "text_lengths", "speech", "speech_lengths")]
    with torch.no_grad():
        (loss, stats, imv, ralpha, mel_pred, _), extra = m._forward_impl(*args, keep=True)
        out = m(*args)                                              # the plain call (fused soft index, graph on the second call)
        out = m(*args)
    torch.cuda.synchronize()
    # (k_size 11 with these random parameters gives |mel| up to 82, 6x the other fixtures': 1e-3 absolute or 2.5e-5 of the largest value)
    mel_tol = max(MEL_TOL, 2.5e-5 * float(np.abs(g["mel_pred"]).max()))
    assert float((mel_pred.cpu() - torch.from_numpy(g["mel_pred"])).abs().max()) <= mel_tol
    assert float((out[4].cpu() - torch.from_numpy(g["mel_pred"])).abs().max()) <= mel_tol
    assert float((ralpha.cpu() - torch.from_numpy(g["reconst_alpha"])).abs().max()) <= 1e-3
    assert float((imv.cpu() - torch.from_numpy(g["imv"])).abs().max()) <= 2e-3
    assert float((extra["dur_pred"].cpu() - torch.from_numpy(g["dur_pred"])).abs().max()) <= 1e-3
    assert float((extra["log_delta_e"].cpu() - torch.from_numpy(g["log_delta_e"])).abs().max()) <= 1e-3
    for got in (loss, out[0]):
        assert abs(float(got) - float(g["loss"])) <= 1e-4 * float(g["loss"])
    assert abs(stats["mel_loss"] - float(g["mel_loss"])) <= 1e-4 * float(g["mel_loss"])
    assert abs(stats["duration_loss"] - float(g["dur_loss"])) <= 2e-4 * max(1.0, float(g["dur_loss"]))


@pytest.mark.parametrize("name", list(VARIANTS))
def test_variant_param_grads_match_reference_golden(golden_dir, name):
    from efficient_tts_amd.train import TrainEngine
    g = np.load(os.path.join(golden_dir, f"variant_{name}.npz"))
    args = [torch.from_numpy(g[k]).to(_dev()) for k in ("text", "text_lengths", "speech", "speech_lengths")]
    eng = TrainEngine(_model(VARIANTS[name]))
    out3, _ = eng.forward_backward(*args)
    torch.cuda.synchronize()
    assert abs(float(out3[0]) - float(g["loss"])) <= 1e-4 * float(g["loss"])
    assert {n for n, _ in eng.layout} == {k[5:] for k in g.files if k.startswith("grad:")}
    for n, _ in eng.layout:
        ref = g["grad:" + n]
        flat = eng.g[n].reshape(-1).cpu().numpy()
        if "grad_stride:" + n in g.files:
            flat = flat[:: int(g["grad_stride:" + n])]
        # fp32 reference gradients vs split-bf16 kernels: 1e-2 of the tensor's max as in test_param_grads_match_reference_golden,
        # except for at most two entries per tensor -- on this 2 x 64-frame batch ONE near-zero (Leaky)ReLU pre-activation whose
        # sign differs under operand rounding moves one channel's gradient by ~3 % of the tensor's max (seen: channel 180 of the
        # duration predictor's first conv with the shared projection, channel 215 of the last mel-encoder layer with the query fc;
        # every other entry of those tensors is within 4e-5 and their norms within 7e-4)
        if n == "text_encoder_key.bias" and np.abs(ref).max() <= 1e-7 * max(100.0, float(g["loss"])):
            # analytically zero (the softmax is shift-invariant) unless the value shares the projection: rounding noise on both sides,
            # growing with the loss (127 / 863 with k_size 7 / 11)
            assert np.abs(flat).max() <= 1e-6 * max(100.0, float(g["loss"])), (n, float(np.abs(flat).max()))
            continue
        scale = max(float(np.abs(ref).max()), 1e-3)
        d = np.abs(flat - ref) / scale
        assert (d > 1e-2).sum() <= 2 and d.max() <= 6e-2, (n, float(d.max()), int((d > 1e-2).sum()))
        gn = float(eng.g[n].double().norm())
        assert abs(gn - float(g["gradnorm:" + n])) <= 5e-3 * float(g["gradnorm:" + n]) + 1e-5, n


@pytest.mark.parametrize("name", ["sharekv", "delta2", "k3", "k7", "k11", "gelu", "elu"])
def test_variant_inference_matches_reference_golden(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"variant_{name}.npz"))
    m = _model(VARIANTS[name])
    ids = torch.from_numpy(g["inf_text"]).to(_dev())
    mel, ralpha = m.inference(ids)
    assert mel.shape[1] == int(g["inf_t2"])
    mel_tol = max(MEL_TOL, 2.5e-5 * float(np.abs(g["inf_mel_pred"]).max()))
    assert float((mel.cpu() - torch.from_numpy(g["inf_mel_pred"])).abs().max()) <= mel_tol
    mels, lens, _ = m.inference_batch(ids, torch.tensor([ids.shape[1]], device=_dev()))
    assert int(lens[0]) == int(g["inf_t2"])
    assert float((mels[0, :int(lens[0])].cpu() - torch.from_numpy(g["inf_mel_pred"])[0]).abs().max()) <= mel_tol


def test_unmasked_loss_sees_nonzero_padding_like_the_reference():
    """use_masking=False with speech that is NOT zero beyond the lengths: the reference's plain means then contain (0 - speech)^2
    on the padded frames (mel_pred is masked to 0 there, fastspeech_loss.py:63-67); checked against the oracle, forward and
    gradients of the training pass (the padded frames must not leak into the head's gradients through the mask, :199-200)."""
    from efficient_tts_amd.train import TrainEngine
    hp = dict(O.DEFAULT_HP, use_masking=False)
    m = _model(dict(use_masking=False))
    g0 = torch.Generator().manual_seed(3)
    text = torch.randint(1, 76, (2, 20), generator=g0)
    mel = torch.randn(2, 90, 80, generator=g0)                     # padding left as noise
    tl, sl = torch.tensor([20, 13]), torch.tensor([90, 61])
    text[1, 13:] = 0
    P = {k: v.clone().requires_grad_(True) for k, v in O.fill_params(hp).items()}
    o = O.forward(P, text, tl, mel, sl, hp)
    o["loss"].backward()
    dev = _dev()
    with torch.no_grad():
        out = m(text.to(dev), tl.to(dev), mel.to(dev), sl.to(dev))
    assert abs(float(out[0]) - float(o["loss"])) <= 1e-4 * float(o["loss"])
    eng = TrainEngine(m)
    out3, _ = eng.forward_backward(text.to(dev), tl.to(dev), mel.to(dev), sl.to(dev))
    torch.cuda.synchronize()
    assert abs(float(out3[0]) - float(o["loss"])) <= 1e-4 * float(o["loss"])
    for n in ("mel_output_layer.weight", "mel_output_layer.bias", "decoder.layers.5.conv.0.bias", "text_encoder_key.weight"):
        ref = P[n].grad
        got = eng.g[n].cpu()
        assert float((got - ref).abs().max()) <= 1e-2 * max(float(ref.abs().max()), 1e-3), n


@pytest.mark.parametrize("k,precision", [(9, "bf16x3"), (1, "bf16x3"), (7, "bf16")])
def test_other_kernel_sizes_match_the_oracle(k, precision):
    """k_size 9 (no fixture: the oracle's restatement is pinned by the reference's k3 / k5 / k7 / k11 fixtures on either side), 1, and 7 on
    plain bf16 operands: forward + free-running inference against the oracle on a ragged batch; the row spaces carry (k - 1) / 2 gap rows"""
    from efficient_tts_amd import EfficientTTSCNN
    hp = dict(O.DEFAULT_HP, k_size=k)
    P = O.fill_params(hp)
    m = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, sigma=0.01, precision=precision, k_size=k)
    assert list(m.state_dict().keys()) == list(P.keys()) and m.row_gap == max(2, (k - 1) // 2)
    m.load_state_dict(P)
    m = m.to(_dev()).eval()
    g0 = torch.Generator().manual_seed(9)
    text = torch.randint(1, 76, (3, 19), generator=g0)
    mel = torch.randn(3, 70, 80, generator=g0)
    tl, sl = torch.tensor([19, 12, 5]), torch.tensor([70, 41, 33])
    for b in range(3):
        text[b, tl[b]:] = 0
        mel[b, sl[b]:] = 0
    ref = O.forward(P, text, tl, mel, sl, hp)
    with torch.no_grad():
        out = m(text.to(_dev()), tl.to(_dev()), mel.to(_dev()), sl.to(_dev()))
    scale = float(ref["mel_pred"].abs().max())
    err = float((out[4].cpu() - ref["mel_pred"]).abs().max())
    if precision == "bf16x3":
        assert err <= max(MEL_TOL, 2.5e-5 * scale), (err, scale)
        assert abs(float(out[0]) - float(ref["loss"])) <= 1e-4 * float(ref["loss"])
        ids = text[:1, :19]
        oi = O.inference(P, ids, hp)
        mel_i, _ = m.inference(ids.to(_dev()))
        assert mel_i.shape[1] == oi["mel_pred"].shape[1]
        assert float((mel_i.cpu() - oi["mel_pred"]).abs().max()) <= max(MEL_TOL, 2.5e-5 * float(oi["mel_pred"].abs().max()))
    else:
        assert err <= 2e-2 * scale, (err, scale)            # bf16 operands: the mode's own error (2^-9 per product), stated, not a parity claim


ACTS = [("Identity", {}), ("ReLU", {}), ("LeakyReLU", {"negative_slope": 0.2, "inplace": True}), ("ELU", {"alpha": 1.3}), ("CELU", {"alpha": 0.8}),
        ("SELU", {}), ("GELU", {}), ("GELU", {"approximate": "tanh"}), ("SiLU", {}), ("Mish", {}), ("Tanh", {}), ("Sigmoid", {}),
        ("Softplus", {"beta": 1.5, "threshold": 4.0}), ("Hardtanh", {"min_val": -0.6, "max_val": 0.9}), ("ReLU6", {}), ("Hardswish", {}),
        ("Hardsigmoid", {}), ("Softsign", {}), ("Tanhshrink", {}), ("LogSigmoid", {})]


@pytest.mark.parametrize("name,params", ACTS, ids=[a + "".join(f"_{k}{v}" for k, v in p.items() if k != "inplace") for a, p in ACTS])
def test_every_supported_activation_matches_the_oracle(name, params):
    """nonlinear_activation = any pointwise torch.nn module the library has a form of (efts_modules.py:32-35: getattr(torch.nn, name)(**params)):
    forward outputs, loss and the parameter gradients of the fused training pass against the oracle (torch's own module, autograd) on a
    ragged batch, bf16x3 operands; 4 layers per stack keep the case small"""
    from efficient_tts_amd import EfficientTTSCNN
    from efficient_tts_amd.train import TrainEngine
    small = dict(n_text_encoder_layer=2, n_mel_encoder_layer=2, n_decoder_layer=2)
    hp = dict(O.DEFAULT_HP, activation=(name, {k: v for k, v in params.items() if k != "inplace"}), **small)
    P = O.fill_params(hp)
    m = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, sigma=0.01, precision="bf16x3", nonlinear_activation=name,
                        nonlinear_activation_params=params, **small)
    assert list(m.state_dict().keys()) == list(P.keys())
    m.load_state_dict(P)
    m = m.to(_dev()).eval()
    g0 = torch.Generator().manual_seed(4)
    text = torch.randint(1, 76, (2, 17), generator=g0)
    mel = torch.randn(2, 60, 80, generator=g0)
    tl, sl = torch.tensor([17, 9]), torch.tensor([60, 37])
    text[1, 9:] = 0
    mel[1, 37:] = 0
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    ref = O.forward(Pg, text, tl, mel, sl, hp)
    ref["loss"].backward()
    args = [t.to(_dev()) for t in (text, tl, mel, sl)]
    with torch.no_grad():
        out = m(*args)
    scale = max(1.0, float(ref["mel_pred"].abs().max()))
    assert float((out[4].cpu() - ref["mel_pred"].detach()).abs().max()) <= max(MEL_TOL, 2.5e-5 * scale)
    assert abs(float(out[0]) - float(ref["loss"])) <= 2e-4 * float(ref["loss"])
    eng = TrainEngine(m)
    out3, _ = eng.forward_backward(*args)
    torch.cuda.synchronize()
    assert abs(float(out3[0]) - float(ref["loss"])) <= 2e-4 * float(ref["loss"])
    for n, _ in eng.layout:
        r = Pg[n].grad
        got = eng.g[n].cpu()
        if n == "text_encoder_key.bias":
            continue                                                # analytically zero: noise on both sides
        sc = max(float(r.abs().max()), 1e-3)
        d = (got - r).abs() / sc
        # On this 2-item batch a pre-activation within operand rounding of a derivative's edge (the duration predictor's ReLU in every case;
        # ReLU / LeakyReLU / ReLU6 / Hardtanh / ELU and SELU with alpha != 1 in the stacks) flips its factor and moves one channel's row of a
        # gradient by a few per cent of the tensor's max (see test_variant_param_grads_match_reference_golden).  A wrong f or f' would move
        # EVERY element by per cents: so the typical element must be within 5e-3 of the tensor's max (a flip upstream reaches everything
        # below it, thinly: seen up to 2.4e-3, smooth activations 1e-5), the tensor as a whole within 5 % in L2, and its norm within 1 %.
        assert float(d.median()) <= 5e-3, (n, float(d.median()))
        assert float((got - r).double().norm()) <= 5e-2 * float(r.double().norm()) + 1e-5, (n, float(d.max()), int((d > 1e-2).sum()))
        assert abs(float(got.double().norm()) - float(r.double().norm())) <= 1e-2 * float(r.double().norm()) + 1e-5, n


def test_unsupported_activations_are_refused_by_name():
    from efficient_tts_amd import EfficientTTSCNN
    for name, params in (("PReLU", {}), ("Softmax", {"dim": 1}), ("GELU", {"approximate": "other"}), ("ELU", {"beta": 1.0})):
        with pytest.raises(NotImplementedError):
            EfficientTTSCNN(num_symbols=76, nonlinear_activation=name, nonlinear_activation_params=params)
