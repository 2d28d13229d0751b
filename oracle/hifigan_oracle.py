"""CPU oracle of the HiFi-GAN V1 generator.

TEST INFRASTRUCTURE ONLY: imported by tests/ and tools/ (golden generation); the product never imports it.

Functional restatement of nntts/vocoders/hifigan_model.py:95-136 (Generator.forward) and :30-58 (ResBlock1)
with the configuration of nntts/vocoders/HiFiGAN_LJ_V1/config.json: upsample rates (8, 8, 2, 2) with kernels
(16, 16, 4, 4), initial 512 channels, residual kernels (3, 7, 11) each with dilations (1, 3, 5).  Parameters are
the reference module's own state_dict (weight_g / weight_v pairs, :100-115), folded here as weight_norm does.
Pinned against the reference itself: tools/gen_golden_hifigan.py imports the reference Generator, fills it with
`fill_params` and stores inputs + outputs (tests/golden/hifigan_*.npz); the shipped checkpoint `generator_v1` is
not in the repository (.MISSING_LARGE_BLOBS), so parity is on seeded weights.
"""
from __future__ import annotations

import zlib
from typing import Dict

import torch
import torch.nn.functional as F

UPSAMPLE_RATES = (8, 8, 2, 2)
UPSAMPLE_KERNELS = (16, 16, 4, 4)
INITIAL_CHANNELS = 512
RES_KERNELS = (3, 7, 11)
RES_DILATIONS = ((1, 3, 5), (1, 3, 5), (1, 3, 5))
LRELU_SLOPE = 0.1


def param_shapes() -> Dict[str, tuple]:
    """state_dict keys and shapes of the reference Generator, in registration order (:100-115)"""
    sh: Dict[str, tuple] = {}

    def wn(prefix, wshape, norm_dim_size):
        sh[prefix + ".bias"] = (wshape[1] if len(wshape) == 3 and prefix.startswith("ups") else wshape[0],)
        sh[prefix + ".weight_g"] = (norm_dim_size, 1, 1)
        sh[prefix + ".weight_v"] = wshape

    wn("conv_pre", (INITIAL_CHANNELS, 80, 7), INITIAL_CHANNELS)
    for i, (u, k) in enumerate(zip(UPSAMPLE_RATES, UPSAMPLE_KERNELS)):
        cin, cout = INITIAL_CHANNELS // 2 ** i, INITIAL_CHANNELS // 2 ** (i + 1)
        wn(f"ups.{i}", (cin, cout, k), cin)                       # ConvTranspose1d weight [cin][cout][k], weight_norm dim 0
    n = 0
    for i in range(len(UPSAMPLE_RATES)):
        ch = INITIAL_CHANNELS // 2 ** (i + 1)
        for k in RES_KERNELS:
            for grp in ("convs1", "convs2"):
                for d in range(3):
                    wn(f"resblocks.{n}.{grp}.{d}", (ch, ch, k), ch)
            n += 1
    wn("conv_post", (1, INITIAL_CHANNELS // 2 ** len(UPSAMPLE_RATES), 7), 1)
    return sh


def fill_params(scale: float = 1.0) -> Dict[str, torch.Tensor]:
    """name-keyed deterministic parameters: direction v ~ N(0,1), gain g chosen so that every layer keeps O(1)
    activations (a variance-preserving fan-in scale), small biases"""
    P = {}
    for name, shape in param_shapes().items():
        g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7fffffff)
        if name.endswith("weight_v"):
            P[name] = torch.randn(*shape, generator=g)
        elif name.endswith("weight_g"):
            vshape = param_shapes()[name[:-1] + "v"]
            if name.startswith("ups"):
                fan = vshape[0] * vshape[2] / UPSAMPLE_RATES[int(name.split(".")[1])]     # taps that hit one output sample
                per = (vshape[1] * vshape[2]) ** 0.5                                    # ||v|| of one dim-0 slice ~ sqrt(cout * k)
            else:
                fan = vshape[1] * vshape[2]
                per = (vshape[1] * vshape[2]) ** 0.5
            P[name] = (scale * per / fan ** 0.5) * (0.8 + 0.4 * torch.rand(*shape, generator=g))
        else:
            P[name] = 0.1 * torch.randn(*shape, generator=g)
    return P


def _fold(P, prefix):
    v, g = P[prefix + ".weight_v"], P[prefix + ".weight_g"]
    return g * v / v.flatten(1).norm(dim=1).view(-1, 1, 1)                              # weight_norm, dim = 0


def forward(P: Dict[str, torch.Tensor], mel: torch.Tensor) -> torch.Tensor:
    """mel [B, 80, T] -> audio [B, 1, T * 256]   (Generator.forward, :117-134)"""
    x = F.conv1d(mel, _fold(P, "conv_pre"), P["conv_pre.bias"], padding=3)                # :118
    n = 0
    for i, (u, k) in enumerate(zip(UPSAMPLE_RATES, UPSAMPLE_KERNELS)):
        x = F.leaky_relu(x, LRELU_SLOPE)                                                 # :120
        x = F.conv_transpose1d(x, _fold(P, f"ups.{i}"), P[f"ups.{i}.bias"], stride=u, padding=(k - u) // 2)   # :121
        xs = None
        for kk in RES_KERNELS:                                                           # :122-127
            r = x
            for d_i, d in enumerate(RES_DILATIONS[0]):                                   # ResBlock1.forward :45-52
                xt = F.leaky_relu(r, LRELU_SLOPE)
                xt = F.conv1d(xt, _fold(P, f"resblocks.{n}.convs1.{d_i}"), P[f"resblocks.{n}.convs1.{d_i}.bias"],
                              padding=(kk * d - d) // 2, dilation=d)
                xt = F.leaky_relu(xt, LRELU_SLOPE)
                xt = F.conv1d(xt, _fold(P, f"resblocks.{n}.convs2.{d_i}"), P[f"resblocks.{n}.convs2.{d_i}.bias"],
                              padding=(kk - 1) // 2)
                r = xt + r
            xs = r if xs is None else xs + r
            n += 1
        x = xs / len(RES_KERNELS)                                                        # :128
    x = F.leaky_relu(x)                                                                  # :129 (default slope 0.01)
    x = F.conv1d(x, _fold(P, "conv_post"), P["conv_post.bias"], padding=3)               # :130
    return torch.tanh(x)                                                                 # :131
