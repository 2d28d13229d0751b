"""CPU oracle for the EFTS-CNN hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A from-scratch, functional restatement (plain torch CPU ops, fp32) of the
algorithm in the reference's ``nntts/models/efficient_tts.py`` and the layers it
calls.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module; the product path
(``efficient_tts_amd``) never does and fails loudly without its HIP library.

Pinning: the reference ships no tests / golden vectors, so
the oracle is pinned against outputs of the reference itself, imported in the
build container by ``tools/gen_golden.py`` and committed as fixtures under
``tests/golden/`` (``tests/test_oracle_golden.py`` checks them on CPU).

Every function cites the reference file:line it follows (paths relative to
/root/reference).  Parameters are passed as a plain ``dict[str, Tensor]`` using
the reference's ``state_dict`` key names, so a reference
checkpoint drops in unchanged.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

# Hyper-parameters of egs/lj/conf/efficient_tts_cnn_phnseq_noDropout.v1.yaml:17-22
# merged over the ctor defaults of nntts/models/efficient_tts.py:26-49.
DEFAULT_HP = dict(
    num_symbols=76,
    odim=80,
    n_channels=512,
    n_text_encoder_layer=5,
    n_mel_encoder_layer=3,
    n_decoder_layer=6,
    n_duration_layer=2,
    k_size=5,
    leaky_slope=0.1,
    activation=None,                       # (torch.nn module name, its keyword arguments) for a nonlinear_activation other than LeakyReLU
                                           # (efts_modules.py:32-35, efficient_tts.py:76-80: getattr(torch.nn, name)(**params)); None: leaky_slope
    duration_offset=1.0,
    sigma=0.01,
    sigma_e=0.5,
    ln_eps=1e-12,
    # ctor options outside the shipped YAML (efficient_tts.py:45-48): the YAML / these defaults select the first value
    use_masking=True,                      # False: FastSpeechLoss means over the padded tensors (fastspeech_loss.py:54-61 skipped)
    share_text_encoder_key_value=False,    # True: value = key projection, no text_encoder_value (:72-75, :150-153)
    use_mel_query_fc=False,                # True: Linear(C, C) on the mel encoder output in front of the attention (:90-95, :163-164)
    delta_e_method_1=True,                 # False: duration target e_{i+1} - e_i with e_{len} = mel length (:205-213); inference from 0 (:261-265)
)


# --------------------------------------------------------------------------
# Deterministic, name-keyed parameter fill (SURVEY.md section 8c recipe).
# The same fill is applied to the imported reference model by
# tools/gen_golden.py, so fixtures need no stored weights.
# --------------------------------------------------------------------------
def _randn(name: str, shape, salt: int = 0) -> torch.Tensor:
    g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + salt) & 0x7FFFFFFF)
    return torch.randn(*shape, generator=g, dtype=torch.float32)


def param_shapes(hp: dict = DEFAULT_HP) -> Dict[str, Tuple[int, ...]]:
    """state_dict keys/shapes of EfficientTTSCNN (nntts/models/efficient_tts.py:57-112)."""
    C, K, V, O = hp["n_channels"], hp["k_size"], hp["num_symbols"], hp["odim"]
    shapes: Dict[str, Tuple[int, ...]] = {"text_embedding_table.weight": (V, C)}

    def block(blk, n):
        for i in range(n):
            p = f"{blk}.layers.{i}.conv.0."
            shapes[p + "bias"] = (C,)
            shapes[p + "weight_g"] = (C, 1, 1)
            shapes[p + "weight_v"] = (C, C, K)

    def linear(name, o, i):
        shapes[name + ".weight"] = (o, i)
        shapes[name + ".bias"] = (o,)

    # module registration order of efficient_tts.py:57-112 (= state_dict order)
    block("text_encoder", hp["n_text_encoder_layer"])
    linear("text_encoder_key", C, C)
    if not hp.get("share_text_encoder_key_value", False):
        linear("text_encoder_value", C, C)
    linear("mel_prenet.0", C, O)
    block("mel_encoder", hp["n_mel_encoder_layer"])
    if hp.get("use_mel_query_fc", False):
        linear("mel_query_fc", C, C)
    block("decoder", hp["n_decoder_layer"])
    linear("mel_output_layer", O, C)
    for i in range(hp["n_duration_layer"]):
        shapes[f"duration_predictor.conv.{i}.0.weight"] = (C, C, 3)
        shapes[f"duration_predictor.conv.{i}.0.bias"] = (C,)
        shapes[f"duration_predictor.conv.{i}.2.weight"] = (C,)
        shapes[f"duration_predictor.conv.{i}.2.bias"] = (C,)
    shapes["duration_predictor.linear.weight"] = (1, C)
    shapes["duration_predictor.linear.bias"] = (1,)
    return shapes


def fill_params(hp: dict = DEFAULT_HP, salt: int = 0) -> Params:
    """Name-keyed deterministic parameters (no dependence on torch's init order)."""
    shapes = param_shapes(hp)
    out: Params = {}
    for name, shp in shapes.items():
        if name.endswith("weight_g"):
            continue
        if name == "text_embedding_table.weight":
            out[name] = _randn(name, shp, salt)
        elif name.endswith(".2.weight"):              # LayerNorm gamma
            out[name] = 1.0 + 0.1 * _randn(name, shp, salt)
        elif name == "duration_predictor.linear.bias":
            # ~6 frames per phoneme so free-running inference yields LJSpeech-like T2
            out[name] = 1.9 + 0.01 * _randn(name, shp, salt)
        elif name.endswith("bias"):
            out[name] = 0.01 * _randn(name, shp, salt)
        else:                                          # conv / linear weights, weight_v
            out[name] = 0.02 * _randn(name, shp, salt)
    for name, shp in shapes.items():
        if name.endswith("weight_g"):
            v = out[name[:-1] + "v"]
            nrm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(shp)
            out[name] = nrm * (1.0 + 0.1 * _randn(name, shp, salt))
    return {k: out[k] for k in shapes}                 # reference key order


# --------------------------------------------------------------------------
# Layers
# --------------------------------------------------------------------------
def non_pad_mask(lengths: torch.Tensor, maxlen: Optional[int] = None) -> torch.Tensor:
    """mask[b,t] = t < len[b]  (nntts/utils/nets_utils.py:58-167,170-254)."""
    if maxlen is None:
        maxlen = int(lengths.max())
    return torch.arange(maxlen)[None, :] < lengths.to(torch.int64)[:, None]


def weight_norm_fold(v: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """w[o] = g[o] * v[o] / ||v[o]||_2, norm over (Cin, k)
    (torch.nn.utils.weight_norm dim=0, applied at nntts/layers/efts_modules.py:92-99)."""
    nrm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, 1, 1)
    return v * (g / nrm)


def conv_weight(P: Params, prefix: str) -> torch.Tensor:
    """Folded conv weight; accepts both the weight-normed (g, v) and the folded
    (`weight`, after remove_weight_norm, efficient_tts.py:400-409) parametrisation."""
    if prefix + "weight" in P:
        return P[prefix + "weight"]
    return weight_norm_fold(P[prefix + "weight_v"], P[prefix + "weight_g"])


def act_of(hp: dict):
    """the stacks' and the prenet's non-linearity: LeakyReLU(leaky_slope) (every reference config), or the torch.nn module the
    reference's ctor would build from (nonlinear_activation, nonlinear_activation_params) (efts_modules.py:32-35)"""
    if hp.get("activation"):
        name, params = hp["activation"]
        return getattr(torch.nn, name)(**params)
    slope = hp["leaky_slope"]
    return lambda t: F.leaky_relu(t, slope)


def res_conv_block(x: torch.Tensor, P: Params, blk: str, n_layers: int, act) -> torch.Tensor:
    """x[B,C,T] -> n x ( x + act(conv1d_k(x)) ), no masking between layers; act = act_of(hp) (or a LeakyReLU slope)
    (nntts/layers/efts_modules.py:48-51,77-79; dropout_rate 0.0 => no Dropout module :30-37)."""
    if not callable(act):
        slope = float(act)
        act = lambda t: F.leaky_relu(t, slope)                                   # noqa: E731
    for i in range(n_layers):
        p = f"{blk}.layers.{i}.conv.0."
        w = conv_weight(P, p)
        x = x + act(F.conv1d(x, w, P[p + "bias"], padding=(w.shape[-1] - 1) // 2))
    return x


def duration_predictor(xs: torch.Tensor, P: Params, n_layers: int, eps: float,
                       pad_mask: Optional[torch.Tensor], inference: bool, offset: float) -> torch.Tensor:
    """xs[B,T1,C] -> log-durations [B,T1]  (nntts/layers/duration_predictor.py:66-88;
    LayerNorm over channels, eps 1e-12: nntts/layers/layer_norm.py:6-30).  Dropout(0.1)
    is the identity here (eval mode / p=0: SURVEY.md section 7 hard part 7)."""
    h = xs.transpose(1, 2)
    for i in range(n_layers):
        p = f"duration_predictor.conv.{i}."
        h = F.relu(F.conv1d(h, P[p + "0.weight"], P[p + "0.bias"], padding=1))
        h = F.layer_norm(h.transpose(1, 2), (h.shape[1],), P[p + "2.weight"], P[p + "2.bias"], eps).transpose(1, 2)
    out = F.linear(h.transpose(1, 2), P["duration_predictor.linear.weight"],
                   P["duration_predictor.linear.bias"]).squeeze(-1)
    if inference:                                   # duration_predictor.py:78-83, to_round=False
        out = torch.clamp(out.exp() - offset, min=0)
    if pad_mask is not None:
        out = out.masked_fill(pad_mask, 0.0)
    return out


# --------------------------------------------------------------------------
# Attention / IMV block
# --------------------------------------------------------------------------
def scaled_dot_attention(q: torch.Tensor, k: torch.Tensor, key_mask: torch.Tensor) -> torch.Tensor:
    """alpha[B,T1,T2] = softmax over T1 of q.k/sqrt(D), padded keys -> 0
    (nntts/models/efficient_tts.py:377-398)."""
    s = torch.bmm(q, k.transpose(1, 2)) / math.sqrt(float(k.shape[-1]))      # [B,T2,T1]
    dead = ~key_mask[:, None, :]
    a = torch.softmax(s.masked_fill(dead, -float("inf")), dim=-1).masked_fill(dead, 0.0)
    return a.transpose(1, 2)


def index_vector(text_mask: torch.Tensor) -> torch.Tensor:
    """p[b,i] = i * text_mask[b,i]  (efficient_tts.py:287-297)."""
    return torch.arange(text_mask.shape[1], dtype=torch.float32)[None, :] * text_mask


def imv_generator(alpha: torch.Tensor, p: torch.Tensor, mel_mask: torch.Tensor,
                  text_lengths: torch.Tensor) -> torch.Tensor:
    """Index-mapping vector pi[B,T2]  (efficient_tts.py:299-324):
    pi'_j = sum_i alpha_ij p_i ; d_j = relu(pi'_j - pi'_{j-1}), d_0 = 0 ;
    pi = cumsum(d) * mel_mask ; pi <- pi / max(max_j pi, 1e-8) * (len_T1 - 1)."""
    soft_idx = torch.einsum("bij,bi->bj", alpha, p)
    d = torch.relu(soft_idx[:, 1:] - soft_idx[:, :-1])
    d = torch.cat([torch.zeros_like(soft_idx[:, :1]), d], dim=1)
    pi = torch.cumsum(d, dim=1) * mel_mask.float()
    top = pi.max(dim=1).values.clamp(min=1e-8)
    return pi / top[:, None] * (text_lengths.float()[:, None] - 1.0)


def aligned_positions(pi: torch.Tensor, p: torch.Tensor, mel_mask: torch.Tensor,
                      text_mask: torch.Tensor, sigma_e: float) -> torch.Tensor:
    """e[B,T1] = sum_j softmax_j(-sigma_e (pi_j - p_i)^2) * q_j, q_j = j*mel_mask_j,
    padded frames excluded from the softmax, padded text rows zeroed
    (efficient_tts.py:326-345)."""
    en = -sigma_e * (pi[:, None, :] - p[:, :, None]) ** 2
    en = en.masked_fill(~mel_mask[:, None, :], -float("inf"))
    beta = torch.softmax(en, dim=2)
    q = torch.arange(mel_mask.shape[1], dtype=torch.float32)[None, :] * mel_mask.float()
    return torch.einsum("bij,bj->bi", beta, q) * text_mask.float()


def reconstruct_alignment(e: torch.Tensor, sigma: float, mel_mask: Optional[torch.Tensor],
                          text_mask: Optional[torch.Tensor], t2: Optional[int] = None) -> torch.Tensor:
    """alpha'[B,T1,T2] = softmax over T1 of -sigma (q_j - e_i)^2; in training q_j = j*mel_mask_j
    (padded columns use q=0) and padded text rows get -inf  (efficient_tts.py:347-375)."""
    if mel_mask is not None:
        t2 = mel_mask.shape[1]
    q = torch.arange(t2, dtype=torch.float32)[None, :].expand(e.shape[0], t2)
    if mel_mask is not None:
        q = q * mel_mask.float()
    en = -sigma * (q[:, None, :] - e[:, :, None]) ** 2
    if text_mask is not None:
        en = en.masked_fill(~text_mask[:, :, None], -float("inf"))
    return torch.softmax(en, dim=1)


# --------------------------------------------------------------------------
# Whole path
# --------------------------------------------------------------------------
def text_side(P: Params, text: torch.Tensor, hp: dict):
    """embed -> text encoder -> key, value  (efficient_tts.py:144-153 / :246-255)."""
    emb = P["text_embedding_table.weight"][text]                                  # [B,T1,C]
    h = res_conv_block(emb.transpose(1, 2), P, "text_encoder", hp["n_text_encoder_layer"],
                       act_of(hp)).transpose(1, 2)
    key = F.linear(h, P["text_encoder_key.weight"], P["text_encoder_key.bias"])
    if hp.get("share_text_encoder_key_value", False):                             # :150-151 / :252-253
        val = key
    else:
        val = F.linear(h, P["text_encoder_value.weight"], P["text_encoder_value.bias"])
    return h, key, val


def decode(P: Params, expanded: torch.Tensor, hp: dict) -> torch.Tensor:
    """[B,C,T2] -> decoder -> mel head [B,T2,odim]  (efficient_tts.py:197-198 / :283-284)."""
    d = res_conv_block(expanded, P, "decoder", hp["n_decoder_layer"], act_of(hp))
    return F.linear(d.transpose(1, 2), P["mel_output_layer.weight"], P["mel_output_layer.bias"])


def forward(P: Params, text: torch.Tensor, text_lengths: torch.Tensor, speech: torch.Tensor,
            speech_lengths: torch.Tensor, hp: dict = DEFAULT_HP, retain: bool = False) -> dict:
    """Teacher-forced forward (efficient_tts.py:120-228).  Returns every intermediate the
    parity tests compare: loss, mel_loss, dur_loss, imv, e, reconst_alpha, mel_pred,
    dur_pred, log_delta_e, alpha (plus text_value / mel_h for kernel-level checks)."""
    text_mask = non_pad_mask(text_lengths, text.shape[1])                       # :137
    mel_mask = non_pad_mask(speech_lengths, speech.shape[1])                    # :139
    both = text_mask[:, :, None] & mel_mask[:, None, :]                         # :141

    _, key, val = text_side(P, text, hp)                                        # :144-153
    key = key * text_mask[:, :, None]                                           # :155-157
    val = val * text_mask[:, :, None]

    pre = act_of(hp)(F.linear(speech, P["mel_prenet.0.weight"], P["mel_prenet.0.bias"]))   # :161 (:76-80)
    mel_h = res_conv_block(pre.transpose(1, 2), P, "mel_encoder", hp["n_mel_encoder_layer"],
                           act_of(hp)).transpose(1, 2)                          # :162
    if hp.get("use_mel_query_fc", False):                                        # :163-164
        mel_h = F.linear(mel_h, P["mel_query_fc.weight"], P["mel_query_fc.bias"])

    alpha = scaled_dot_attention(mel_h, key, text_mask).masked_fill(~both, 0.0)  # :167-168
    p = index_vector(text_mask)                                                  # :171
    imv = imv_generator(alpha, p, mel_mask, text_lengths)                        # :174
    e = aligned_positions(imv, p, mel_mask, text_mask, hp["sigma_e"])            # :178-180
    ralpha = reconstruct_alignment(e, hp["sigma"], mel_mask, text_mask).masked_fill(~both, 0.0)  # :184-186

    expanded = torch.bmm(val.transpose(1, 2), ralpha) * mel_mask[:, None, :]    # :190-194
    if retain:                                   # gradient checks of the hand-written backward
        for t in (expanded, ralpha, e, imv, mel_h, key, val):
            if t.requires_grad:
                t.retain_grad()
    mel_pred = decode(P, expanded, hp) * mel_mask[:, :, None]                   # :197-200

    if hp.get("delta_e_method_1", True):
        delta_e = torch.cat([e[:, :1], e[:, 1:] - e[:, :-1]], dim=1).detach()   # :204 (method 1)
    else:                                                                       # :205-213
        ee = torch.cat([e.detach(), torch.zeros(e.shape[0], 1)], dim=1)
        for i in range(e.shape[0]):
            ee[i, int(text_lengths[i])] = float(speech_lengths[i])
        delta_e = ee[:, 1:] - ee[:, :-1]
    log_delta_e = torch.log(delta_e + hp["duration_offset"]).masked_fill(~text_mask, 0.0)  # :215-216
    dur_pred = duration_predictor(val, P, hp["n_duration_layer"], hp["ln_eps"], ~text_mask,
                                  False, hp["duration_offset"])                 # :219

    if hp.get("use_masking", True):
        # FastSpeechLoss, use_masking=True (nntts/losses/fastspeech_loss.py:54-67)
        n_mel = mel_mask.sum() * speech.shape[2]
        mel_loss = (((mel_pred - speech) ** 2) * mel_mask[:, :, None]).sum() / n_mel
        dur_loss = ((dur_pred - log_delta_e).abs() * text_mask).sum() / text_mask.sum()
    else:
        # use_masking=False (the ctor default, efficient_tts.py:43): plain means over the padded tensors (:63-67)
        mel_loss = ((mel_pred - speech) ** 2).mean()
        dur_loss = (dur_pred - log_delta_e).abs().mean()
    return dict(loss=mel_loss + dur_loss, mel_loss=mel_loss, dur_loss=dur_loss, imv=imv, e=e,
                reconst_alpha=ralpha, mel_pred=mel_pred, dur_pred=dur_pred,
                log_delta_e=log_delta_e, alpha=alpha, text_value=val, text_key=key, mel_h=mel_h,
                expanded=expanded)


def inference(P: Params, text: torch.Tensor, hp: dict = DEFAULT_HP,
              forced_delta: Optional[torch.Tensor] = None) -> dict:
    """Free-running synthesis, B == 1, no masks (efficient_tts.py:230-285).
    `forced_delta` replaces the predicted durations (bench config 2 (ii))."""
    _, _, val = text_side(P, text, hp)                                          # :246-255
    delta = duration_predictor(val, P, hp["n_duration_layer"], hp["ln_eps"], None, True,
                               hp["duration_offset"])                           # :258
    if forced_delta is not None:
        delta = forced_delta
    e = torch.cumsum(delta, dim=1)                                              # :260
    t2 = int(torch.round(e[:, -1]).reshape(-1)[0].item())                        # :361
    if not hp.get("delta_e_method_1", True):                                     # :261-265 + trim_e (:362-363): positions start at 0
        e = e - delta
    ralpha = reconstruct_alignment(e, hp["sigma"], None, None, t2)              # :270-274
    mel_pred = decode(P, torch.bmm(val.transpose(1, 2), ralpha), hp)            # :278-284
    return dict(mel_pred=mel_pred, reconst_alpha=ralpha, delta=delta, e=e, t2=t2)


# --------------------------------------------------------------------------
# Training-step pieces the trainer parity tests need
# --------------------------------------------------------------------------
def warmup_lr(base_lr: float, step_num: int, warmup_steps: int) -> float:
    """lr * w^0.5 * min(s^-0.5, s * w^-1.5)  (nntts/schedulers/warmup_lr.py:44-51)."""
    return base_lr * warmup_steps ** 0.5 * min(step_num ** -0.5, step_num * warmup_steps ** -1.5)


def adam_amsgrad_step(p, g, m, v, vmax, step: int, lr: float, b1=0.9, b2=0.99, eps=1e-9, wd=1e-5):
    """One torch.optim.Adam(amsgrad=True, coupled L2) update, in place
    (egs/lj/conf/efficient_tts_cnn_phnseq_noDropout.v1.yaml:34-40)."""
    g = g + wd * p
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    torch.maximum(vmax, v, out=vmax)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (vmax.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
